#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels in one .hip file (VALU / SALU / LDS / VMEM / waitcnt counts).

Usage: python tools/isa_mix.py <file.hip> [demangled-name filter ...]
"""
import collections
import re
import subprocess
import sys
import tempfile

src, filters = sys.argv[1], sys.argv[2:]
with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only",
                    "-Iinclude", "-S", src, "-o", tmp.name], check=True, capture_output=True)
    lines = open(tmp.name).read().split("\n")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    i += 1
    if not m:
        continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    mix = collections.Counter()
    while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
        text = lines[i].strip()
        i += 1
        if not text or text[0] in ".;/" or text.endswith(":"):
            continue
        op = text.split()[0]
        kind = ("wait" if op.startswith("s_waitcnt") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_")
                else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        mix[kind] += 1
    if all(f in name for f in filters):
        print(f"{name[:120]}\n    total {sum(mix.values())}  " + "  ".join(f"{k} {v}" for k, v in sorted(mix.items())))
