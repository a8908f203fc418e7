#!/usr/bin/env python3
"""Which 16-bit ingredient of the bf16 mode costs volume PSNR at convergence?  chest_50's own schedule (bench.full_schedule) with the
table storage type and the MLP operand type chosen independently; every run is scored twice: from the fp32 master parameters in the
fp32 parity mode (what bench.py / train.py evaluate) and through the arithmetic the run trained with.
    python tools/precision_grid.py [--epochs 1000] [--out gpurun_out/precision_grid.jsonl]"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd import _abi, fused  # noqa: E402

DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
MP = {"fp32": _abi.F32, "bf16": _abi.BF16}


def psnr_as_trained(scan, engine):
    """Volume PSNR through the table the kernels read (the 16-bit shadow) and the MLP operand type of the training steps."""
    net = engine.net
    starts, stops, dims = scan.axes
    dims = [int(v) for v in dims]
    cfg = fused.render_cfg(net, 2, False, engine.mlp_precision, forward_only=True)
    cfg.table_dtype = _abi.dtype_code(engine.table_dtype)
    B = dims[0] * dims[1] * dims[2]
    ws = fused.forward_workspace(cfg, B, engine.device)
    sigma = torch.empty(dims, device=engine.device, dtype=torch.float32)
    a, b, d = (ctypes.c_double * 3)(*[float(v) for v in starts]), (ctypes.c_double * 3)(*[float(v) for v in stops]), (ctypes.c_uint32 * 3)(*dims)
    _abi.check(_abi.lib().naf_field_forward_grid(ctypes.byref(a), ctypes.byref(b), ctypes.byref(d), _abi.ptr(engine.table), _abi.ptr(engine.offsets),
                                                 _abi.ptr(engine.mlp), _abi.ptr(sigma), ctypes.byref(cfg), _abi.ptr(ws), ws.numel(), _abi.stream_ptr()),
               "field_forward_grid")
    mse = float((sigma.double() - scan.image).square().mean().item())
    return 20.0 * float(np.log10(1.0 / np.sqrt(mse)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--combos", default="bf16:bf16,fp32:bf16,bf16:fp32,fp16:bf16,fp32:fp32")
    ap.add_argument("--seeds", default="0", help="comma-separated seeds: parameter initialisation, jitter AND pixel draws change with the seed")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "precision_grid.jsonl"))
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    scan = bench.ChestScan(device, 1234)
    n_rays = bench.CHEST["yaml_rays"]
    rays = torch.empty(n_rays, 8, device=device)
    weight, loss_name = bench.step_weights(n_rays, device)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    for combo, seed in [(c, int(sd)) for sd in args.seeds.split(",") for c in args.combos.split(",")]:
        table, mlp = combo.split(":")
        scan.sampler.seed = 1234 + 1000 * seed
        from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder
        from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
        from neuralvolumetricreconstructionformedicalimages_amd.network import DensityNetwork
        torch.manual_seed(seed)
        C = bench.CHEST
        enc = HashEncoder(3, C["num_levels"], C["level_dim"], C["base_resolution"], C["log2_hashmap_size"])
        net = DensityNetwork(enc, bound=C["bound"], num_layers=4, hidden_dim=32, skips=[2], out_dim=1, last_activation="sigmoid").to(device)
        engine = NAFEngine(net, C["n_samples"], perturb=True, lr=C["lr"], table_dtype=DT[table], mlp_precision=MP[mlp], seed=seed)
        t_train, step, curve = 0.0, 0, []
        for e0 in range(0, args.epochs, 100):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(min(100, args.epochs - e0) * scan.raygen.n_projections):
                target, _r = scan.sampler.draw(step, n_rays, rays)
                engine.train_step(rays, target, weight, ray_base=step * n_rays)
                step += 1
            torch.cuda.synchronize()
            t_train += time.perf_counter() - t0
            curve.append(round(scan.volume_psnr(engine.net), 3))
        rec = {"table": table, "mlp": mlp, "seed": seed, "epochs": args.epochs, "steps": step, "train_seconds": round(t_train, 2), "loss": loss_name,
               "psnr_fp32_master_fp32_eval": round(scan.volume_psnr(engine.net), 3), "psnr_as_trained": round(psnr_as_trained(scan, engine), 3),
               "psnr_every_100_epochs": curve}
        print(json.dumps(rec), flush=True)
        with open(args.out, "a") as f:
            f.write(json.dumps(rec) + "\n")
        del engine, net, enc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
