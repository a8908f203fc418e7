#!/usr/bin/env python3
"""Training-step time of the fused engine for arbitrary NAF shapes (bench.py is pinned to chest_50, BASELINE.json configs[1]):

    python tools/step_bench.py --log2T 22 --samples 320 --table fp16 --rays 32768        # foot_50 shapes (configs[4])
    python tools/step_bench.py --log2T 19 --samples 576 --table bf16 --rays 16384        # abdomen_50 shapes (configs[3])

Rays cross the volume like cone-beam rays of the synthetic scan; targets are random (throughput only).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralvolumetricreconstructionformedicalimages_amd import _abi  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.network import DensityNetwork  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2T", type=int, default=19)
ap.add_argument("--samples", type=int, default=192)
ap.add_argument("--table", choices=["fp32", "bf16", "fp16"], default="bf16")
ap.add_argument("--rays", type=int, default=65536)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--flags", type=int, default=0, help="naf_render_cfg.flags (diagnostics, e.g. 32 = NAF_CFG_ENCODE_TWO_GATHERS)")
args = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
net = DensityNetwork(HashEncoder(3, 16, 2, 16, args.log2T), bound=0.3, num_layers=4, hidden_dim=32, skips=[2], out_dim=1,
                     last_activation="sigmoid").to(dev)
tdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.table]
engine = NAFEngine(net, args.samples, perturb=True, lr=1e-3, table_dtype=tdt, cfg_flags=args.flags)
n = args.rays
ang = torch.rand(n, device=dev) * 6.283
o = torch.stack([torch.cos(ang), torch.sin(ang), (torch.rand(n, device=dev) - 0.5) * 0.2], -1)
tgt = (torch.rand(n, 3, device=dev) - 0.5) * 0.25
d = tgt - o
rays = torch.cat([o, d, torch.full((n, 1), 0.814, device=dev), torch.full((n, 1), 1.186, device=dev)], -1).contiguous()
target = torch.rand(n, device=dev) * 0.1
weight = torch.full((n,), 1.0 / n, device=dev)
for _ in range(2):
    engine.train_step(rays, target, weight)
torch.cuda.synchronize()
_abi.profile_enable(True)
t0 = time.perf_counter()
for _ in range(args.steps):
    engine.train_step(rays, target, weight)
torch.cuda.synchronize()
el = time.perf_counter() - t0
prof = _abi.profile_collect()
_abi.profile_enable(False)
print(json.dumps({"flags": args.flags, "log2T": args.log2T, "samples": args.samples, "table": args.table, "rays_per_step": n,
                  "ms_per_step": round(el / args.steps * 1e3, 3), "rays_per_s": n * args.steps / el,
                  "table_MB": round(net.encoder.embeddings.numel() * {"fp32": 4, "bf16": 2, "fp16": 2}[args.table] / 1e6, 1),
                  "kernels_ms_per_step": {k: round(ms / args.steps, 3) for k, (c, ms) in sorted(prof.items())}}))
