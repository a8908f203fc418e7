#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel.

Usage: python tools/kernel_resources.py <file.hip> [filter]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem)
    if flt and flt not in dem:
        continue
    print(f"{dem[:110]:110s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>4} spill {r.get('VGPR Spill', r.get('VGPRs Spill','?')):>3} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>5} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')}")
