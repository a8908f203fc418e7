#!/usr/bin/env python3
"""Does a per-step working set below the 256 MiB Infinity Cache run faster than one above it?

A 1 024-ray training step touches ~376 MB (Adam state 171 MB read + written, 16-bit table 28.5 MB, scatter records 151 MB written then
read, features 25 MB) in the same order every step: a cyclic sweep larger than the cache.  This probe times cyclic sweeps of
`a[i] = a[i] * 0.999 + 1` (read + write, like the optimiser pass) over buffers of 64 .. 512 MB, and the same with a second buffer
that is written once and read once between two sweeps (like the records), and prints bytes moved per second.

    python tools/mall_probe.py
"""
import json
import time

import torch

dev = torch.device("cuda")


def sweep_rate(n_bytes, extra_bytes=0, reps=30):
    a = torch.zeros(n_bytes // 4, device=dev)
    e = torch.zeros(max(extra_bytes // 4, 1), device=dev)
    s = torch.zeros(max(extra_bytes // 4, 1), device=dev)

    def one():
        torch.add(a, 1.0, out=a)          # one kernel: read + write of the whole buffer
        if extra_bytes:
            e.fill_(1.0)                  # written once ...
            torch.add(e, 1.0, out=s)      # ... read once (and a second stream written)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    moved = 2 * n_bytes + (3 * extra_bytes if extra_bytes else 0)
    return dt, moved / dt / 1e12


rows = []
for mb in (64, 128, 171, 200, 232, 256, 300, 376, 512, 1024):
    dt, rate = sweep_rate(mb << 20)
    rows.append({"sweep_MB": mb, "extra_MB": 0, "ms": round(dt * 1e3, 4), "TB_per_s": round(rate, 2)})
    print(json.dumps(rows[-1]), flush=True)
for mb, extra in ((200, 16), (200, 40), (200, 80), (200, 151), (171, 151), (171, 40)):
    dt, rate = sweep_rate(mb << 20, extra << 20)
    rows.append({"sweep_MB": mb, "extra_MB": extra, "ms": round(dt * 1e3, 4), "TB_per_s": round(rate, 2)})
    print(json.dumps(rows[-1]), flush=True)
