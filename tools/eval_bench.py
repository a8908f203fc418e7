#!/usr/bin/env python3
"""Inference-side throughput of the fused field (the evaluation half of SURVEY 8f-2): the 256^3 volume query
(train.py:246-250: run_network(voxels)) and one full 512x512 projection render (train.py:237), chest_50 shapes.

    python tools/eval_bench.py [--precision bf16|fp32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralvolumetricreconstructionformedicalimages_amd import fused, phantom  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.geometry import ConeGeometry, RayGenerator, get_voxels  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.network import DensityNetwork  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--fused", action="store_true", help="the single fused forward kernel (NAF_CFG_FORWARD_FUSED, opt-in) instead of encode_kernel + mlp_forward_kernel")
ap.add_argument("--two-kernel", action="store_true", help="(the default since round 3; kept for old command lines)")
ap.add_argument("--two-gathers", action="store_true", help="NAF_CFG_ENCODE_TWO_GATHERS (two-kernel path only)")
ap.add_argument("--store-features", action="store_true",
                help="NAF_CFG_FUSED_STORE_FEATURES: the fused kernel also writes the [L, B, C] features (what the feature traffic alone costs it)")
args = ap.parse_args()
dev = torch.device("cuda")
fused.forward_fused = bool(args.fused) and not args.two_kernel
if args.two_gathers:
    fused._default_flags |= 32
if args.store_features:
    fused._default_flags |= 16
geo = ConeGeometry(phantom.scan_geometry(256, "cone"))
torch.manual_seed(0)
net = DensityNetwork(HashEncoder(3, 16, 2, 16, 19), bound=0.3, num_layers=4, hidden_dim=32, skips=[2], out_dim=1,
                     last_activation="sigmoid").to(dev)
if args.precision == "bf16":
    net.encoder.embeddings.data = net.encoder.embeddings.data.to(torch.bfloat16)
voxels = torch.tensor(get_voxels(geo), dtype=torch.float32, device=dev)          # [256,256,256,3]
raygen = RayGenerator(geo, np.linspace(0, np.pi, 51)[:-1], dev)
rays = raygen.rays_for_projection(7)                                             # [512*512, 8]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.iters


with torch.no_grad():
    t_vol = timed(lambda: fused.field_query(net, voxels))
    sv = geo.sVoxel / 2 - geo.dVoxel / 2
    t_grid = timed(lambda: fused.field_query_grid(net, [-float(v) for v in sv], [float(v) for v in sv], [256, 256, 256]))
    t_proj = timed(lambda: fused.fused_render(rays, net, 192, False))
print(json.dumps({"precision": args.precision, "forward": "fused kernel" if fused.forward_fused and args.precision != "fp32" else "two kernels",
                  "two_gathers": args.two_gathers, "store_features": args.store_features,
                  "volume_query_256^3": {"points": voxels.numel() // 3, "ms": round(t_vol * 1e3, 3), "points_per_s": voxels.numel() / 3 / t_vol},
                  "volume_query_256^3_generated_grid": {"ms": round(t_grid * 1e3, 3), "points_per_s": voxels.numel() / 3 / t_grid},
                  "projection_512x512_S192": {"rays": rays.shape[0], "ms": round(t_proj * 1e3, 3), "rays_per_s": rays.shape[0] / t_proj}}))
