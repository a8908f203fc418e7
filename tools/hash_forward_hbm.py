#!/usr/bin/env python3
"""Stand-alone hash-encoder forward (naf_hash_encode_forward, the reference ABI) on a table far larger than the
256 MiB Infinity Cache, so that the gathers are served by HBM: reports algorithmic GB/s; run it under
`rocprofv3 --pmc FETCH_SIZE` for the measured HBM bytes.

    python tools/hash_forward_hbm.py --log2T 24 --points 4194304 --dtype float32
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralvolumetricreconstructionformedicalimages_amd import _abi  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.encoder import level_offsets  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2T", type=int, default=24)
ap.add_argument("--points", type=int, default=1 << 22)
ap.add_argument("--dtype", default="float32")
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dt = getattr(torch, args.dtype)
offs = torch.from_numpy(level_offsets(3, 16, 16, args.log2T)).cuda()
rows = int(offs[-1])
emb = (torch.rand(rows, 2, device="cuda") - 0.5).to(dt)
x = torch.rand(args.points, 3, device="cuda")
out = torch.empty(16, args.points, 2, device="cuda", dtype=dt)


def run():
    _abi.check(_abi.lib().naf_hash_encode_forward(_abi.ptr(x), _abi.ptr(emb), _abi.ptr(offs), _abi.ptr(out), args.points, 3, 2, 16, 16, 0,
                                                  None, _abi.dtype_code(dt), _abi.LAYOUT_LBC, _abi.stream_ptr()))


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    run()
torch.cuda.synchronize()
dt_s = (time.perf_counter() - t0) / args.iters
es = emb.element_size()
alg = args.points * (12 + 16 * 8 * 2 * es + 32 * es)
print(json.dumps({"log2T": args.log2T, "table_MB": rows * 2 * es / 1e6, "points": args.points, "dtype": args.dtype, "ms": dt_s * 1e3,
                  "points_per_s": args.points / dt_s, "algorithmic_GBps": alg / dt_s / 1e9, "gathers_per_s": args.points * 128 / dt_s}))
