#!/usr/bin/env python3
"""Train the synthetic chest_50 scan (256^3 phantom, 50 cone-beam projections of 512x512) with the fused engine and
report reconstruction PSNR (get_psnr_3d, the metric of the +-0.1 dB bar) against wall-clock time.

    python tools/train_chest.py --rays 16384 --steps 3000 --precision bf16 --eval-every 500
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from neuralvolumetricreconstructionformedicalimages_amd import phantom  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.geometry import ConeGeometry, RayGenerator, get_voxels  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.network import DensityNetwork  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.utils import get_psnr_3d  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--n-voxel", type=int, default=256)
    ap.add_argument("--precision", choices=["bf16", "fp16", "fp32"], default="bf16", help="table storage (fp32 = parity mode)")
    ap.add_argument("--log2T", type=int, default=19, help="log2 of the hash-table rows per level (chest_50: 19, foot_50: 22)")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--eval-every", type=int, default=500)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda")
    geo = ConeGeometry(phantom.scan_geometry(args.n_voxel, "cone"))
    angles = np.linspace(0, np.pi, 51)[:-1]
    raygen = RayGenerator(geo, angles, dev)
    table = phantom.ellipsoid_table(seed=0, extent=float(geo.sVoxel[0]) / 2)
    image = phantom.volume(geo, table, device=dev)
    voxels = torch.tensor(get_voxels(geo), dtype=torch.float32, device=dev)
    # projections of the whole scan (the "dataset"): 50 x 512 x 512 analytic line integrals
    projs = torch.cat([phantom.line_integrals(raygen.rays_for_projection(i), table) for i in range(len(angles))])
    torch.manual_seed(0)
    enc = HashEncoder(3, 16, 2, 16, args.log2T)
    net = DensityNetwork(enc, bound=0.3, num_layers=4, hidden_dim=32, skips=[2], out_dim=1, last_activation="sigmoid").to(dev)
    engine = NAFEngine(net, 192, perturb=True, lr=args.lr, table_dtype={"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.precision])
    n_pix = projs.numel()
    valid = torch.nonzero(projs.abs() > 0).reshape(-1)              # tigre.py:356: only pixels that saw the object
    weight = torch.full((args.rays,), 1.0 / args.rays, device=dev)
    rays = torch.empty(args.rays, 8, device=dev)
    log = []
    t_train = 0.0

    def evaluate(step):
        with torch.no_grad():
            vol = torch.cat([net(voxels[i:i + 32].reshape(-1, 3)).reshape(-1, args.n_voxel, args.n_voxel)
                             for i in range(0, args.n_voxel, 32)])
        psnr = float(get_psnr_3d(vol, image))
        entry = {"step": step, "rays": step * args.rays, "train_seconds": round(t_train, 3), "psnr_3d_db": round(psnr, 3),
                 "loss": float(engine.loss.item())}
        log.append(entry)
        print(json.dumps(entry), flush=True)

    evaluate(0)
    for step in range(1, args.steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pix = valid[torch.randint(0, valid.numel(), (args.rays,), device=dev)]
        raygen.rays_for_pixels(pix, out=rays)
        engine.train_step(rays, projs[pix], weight, ray_base=step * args.rays)
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t0
        if step % args.eval_every == 0 or step == args.steps:
            evaluate(step)
    if args.out:
        json.dump({"args": vars(args), "log": log}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
