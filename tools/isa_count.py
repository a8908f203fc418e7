#!/usr/bin/env python3
"""Instruction census of one kernel in a gfx950 .s file (hipcc -save-temps): per label block, counts by class.
usage: isa_count.py file.s kernel-name-substring [--blocks]"""
import collections
import re
import sys

def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"

def main():
    path, name = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and re.match(r"^_Z\S+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ("entry", collections.Counter(), collections.Counter())
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), collections.Counter(), collections.Counter())
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        op = t.split()[0]
        cur[1][klass(op)] += 1
        cur[2][op] += 1
    blocks.append(cur)
    total, ops = collections.Counter(), collections.Counter()
    for _, c, o in blocks:
        total.update(c); ops.update(o)
    print("kernel total:", dict(total))
    if show_blocks:
        for n, c, o in blocks:
            if sum(c.values()) >= 12:
                print(f"{n:14s} {dict(c)}")
    print("top ops:", ops.most_common(45))

main()
