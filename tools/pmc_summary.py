#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch.

Usage: python tools/pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv [--json out.json]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B?  On gfx950 the guide
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section) says: bytes = value * 1024; FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads (x2 correction), uncalibrated for other widths.  We report raw*1024 and the
x2-corrected read figure side by side.
"""
import csv
import json
import re
import sys
from collections import defaultdict

args = sys.argv[1:]
out_json = None
if "--json" in args:
    i = args.index("--json")
    out_json = args[i + 1]
    del args[i:i + 2]
files = args
agg = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"<.*", "", name).replace("naf::", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in sorted(agg.items()):
    if not any(s in k for s in ("encode", "scatter", "mlp", "adam", "hash_", "fused_forward", "draw_scan")):
        continue
    row = {c: sum(v) / len(v) * (1024 if c.endswith("_SIZE") else 1) for c, v in cs.items()}     # *_SIZE are in KiB, the rest are counts
    row["dispatches"] = max(len(v) for v in cs.values())
    summary[k] = row
    print(f"{k:28s} n={row['dispatches']:3d} " + " ".join(f"{c}={row[c]/1e6:10.1f} {'MB' if c.endswith('_SIZE') else 'M'}/dispatch" for c in cs))
if out_json:
    json.dump(summary, open(out_json, "w"), indent=1)
