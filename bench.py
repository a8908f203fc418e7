#!/usr/bin/env python3
"""bench.py -- training rays/s of the NAF hot path on chest_50 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = ray generation for one batch of pixels + fused render forward + masked MSE + backward (MLP + hash-table
scatter) + gradient all-reduce (N > 1) + Adam over the table and the MLP, i.e. `Trainer.train_step` of the reference
(src/trainer.py:134-142 around train.py:48-135) on chest_50.yaml: 256^3 volume, 50 cone-beam projections of
512x512, L=16 T=2^19 C=2 hash grid stored in bf16, S=192 samples per ray, synthetic phantom with analytic
projections (the reference ships no data).  Inputs (poses, pixel indices, targets) are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CHEST = dict(n_voxel=256, n_proj=50, n_samples=192, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
             bound=0.3, lr=1e-3)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_point(kernel, table_bytes, feat_bytes, L=16, C=2, D=3):
    """hash fwd = D*4 + L*2^D*C*s_t + L*C*s_o ; hash bwd = L*C*s_g + D*4 + L*2^D*C*4*2 (fp32 atomic RMW).
    In the fused pipeline the sample position is recomputed from the 32-byte ray record, so the D*4 term is
    replaced by 32 B per ray / S samples (negligible); we keep the SURVEY formula for comparability."""
    if kernel == "encode_kernel":
        return D * 4 + L * (2 ** D) * C * table_bytes + L * C * feat_bytes
    if kernel == "hash_backward_kernel":
        return L * C * feat_bytes + D * 4 + L * (2 ** D) * C * 4 * 2
    raise KeyError(kernel)


def cpu_baseline(seconds, n_rays, seed=0):
    """The oracle's pure-PyTorch CPU training step (the reference has no CPU hash encoder: SURVEY.md 8c/8d),
    same chest_50 shapes, fp32, all host threads.  Bounded: warm-up + as many steps as fit in `seconds`."""
    from oracle.hashgrid_ref import HashEncoderRef
    from oracle.network_ref import DensityNetworkRef
    from oracle import render_ref as R

    torch.manual_seed(seed)
    threads = min(len(os.sched_getaffinity(0)), 32)     # the GPU box hands this job a CPU share, not the whole host
    torch.set_num_threads(threads)
    enc = HashEncoderRef(3, CHEST["num_levels"], CHEST["level_dim"], CHEST["base_resolution"], CHEST["log2_hashmap_size"])
    net = DensityNetworkRef(enc, bound=CHEST["bound"], num_layers=4, hidden_dim=32, skips=(2,), out_dim=1)
    opt = torch.optim.Adam(net.parameters(), lr=CHEST["lr"], betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(seed)
    ang = torch.rand(n_rays, generator=g) * 3.1416
    o = torch.stack([torch.cos(ang), torch.sin(ang), torch.zeros(n_rays)], -1)
    tgt = (torch.rand(n_rays, 3, generator=g) - 0.5) * 0.25
    d = tgt - o
    rays = torch.cat([o, d, torch.full((n_rays, 1), 0.814), torch.full((n_rays, 1), 1.186)], -1)
    target = torch.rand(n_rays, generator=g) * 0.1

    def step():
        opt.zero_grad()
        acc = R.render(rays, net, None, CHEST["n_samples"], 0, True, 409600, 0.0)["acc"]
        loss = ((acc - target) ** 2).mean()
        loss.backward()
        opt.step()

    full = (rays, target)
    rays, target = rays[:64], target[:64]
    step()                                     # warm-up on a small batch (pages the table in, builds autograd caches)
    rays, target = full
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": n * n_rays / el, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{n} optimiser steps of {n_rays} rays x {CHEST['n_samples']} samples (chest_50 shapes, fp32, "
                      f"oracle/ pure-PyTorch path, torch {torch.__version__}) in {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=65536, help="rays per GPU per step (chest_50.yaml uses 1024)")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16", help="table storage / MLP operand type")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 disables it)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scatter-mode", choices=["auto", "atomic", "binned"], default="auto",
                    help="table-gradient scatter: auto = binned from 2^13 points per step on (naf_set_scatter_mode)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the ray batch is pipelined over (engine n_streams)")
    ap.add_argument("--chunk-rays", type=int, default=16384, help="rays per pipelined chunk when --streams > 1")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    # Rehearsal hooks for a one-GPU box (the real multi-GPU run uses neither): NAF_BENCH_BACKEND=gloo swaps RCCL for gloo,
    # NAF_BENCH_SHARE_GPU=1 puts every rank on device 0.
    backend = os.environ.get("NAF_BENCH_BACKEND", "nccl")
    if os.environ.get("NAF_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    from neuralvolumetricreconstructionformedicalimages_amd import _abi, phantom
    from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    from neuralvolumetricreconstructionformedicalimages_amd.geometry import ConeGeometry, RayGenerator
    from neuralvolumetricreconstructionformedicalimages_amd.network import DensityNetwork

    _abi.check(_abi.lib().naf_set_scatter_mode({"auto": 0, "atomic": 1, "binned": 2}[args.scatter_mode]), "set_scatter_mode")

    # ---- chest_50 scan: geometry, poses, phantom ----------------------------------------------------------------
    geo = ConeGeometry(phantom.scan_geometry(CHEST["n_voxel"], "cone"))
    angles = np.linspace(0, np.pi, CHEST["n_proj"] + 1)[:-1]          # generateData.py:175, totalAngle 180
    raygen = RayGenerator(geo, angles, device)
    table = phantom.ellipsoid_table(seed=0, extent=float(geo.sVoxel[0]) / 2)

    torch.manual_seed(args.seed)                                       # identical initial weights on every rank
    enc = HashEncoder(3, CHEST["num_levels"], CHEST["level_dim"], CHEST["base_resolution"], CHEST["log2_hashmap_size"])
    net = DensityNetwork(enc, bound=CHEST["bound"], num_layers=4, hidden_dim=32, skips=[2], out_dim=1,
                         last_activation="sigmoid").to(device)
    tdt = torch.bfloat16 if args.precision == "bf16" else torch.float32
    engine = NAFEngine(net, CHEST["n_samples"], perturb=True, lr=CHEST["lr"], table_dtype=tdt, seed=args.seed,
                       process_group=pg, n_streams=args.streams, chunk_rays=args.chunk_rays)

    # ---- per-step inputs, resident in HBM before the clock starts ------------------------------------------------
    n = args.rays
    total_steps = args.warmup + args.steps
    gen = torch.Generator(device=device).manual_seed(1234 + rank)      # each rank draws its own ray shard
    n_pix = raygen.n_projections * raygen.pixels_per_projection
    pixels = torch.randint(0, n_pix, (total_steps, n), device=device, generator=gen)
    targets = torch.empty(total_steps, n, device=device)
    rays = torch.empty(n, 8, device=device)
    for i in range(total_steps):
        raygen.rays_for_pixels(pixels[i], out=rays)
        for j in range(0, n, 1 << 16):
            targets[i, j:j + (1 << 16)] = phantom.line_integrals(rays[j:j + (1 << 16)], table)
    weight = torch.full((n,), 1.0 / (n * world), device=device)       # global mean over all ranks' rays

    def step(i):
        raygen.rays_for_pixels(pixels[i], out=rays)                   # G3: on-the-fly cone-beam ray generation
        return engine.train_step(rays, targets[i], weight, ray_base=(i * world + rank) * n)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"inputs resident: {total_steps} x {n} rays; starting {args.warmup} warm-up steps")
    for i in range(args.warmup):
        step(i)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    barrier()
    log(f"timing {args.steps} steps")
    _abi.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        loss = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = _abi.profile_collect()
    _abi.profile_enable(False)
    log(f"timed region {elapsed:.3f} s")
    final_loss = float(loss.item())
    overflow = engine.scatter_overflow(n)

    allreduce_ms = None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        try:        # informational: cost of the one collective of a step (table + MLP gradients + loss), outside the timed region
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            engine.grad_flat.zero_()
            torch.distributed.all_reduce(engine.grad_flat)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                torch.distributed.all_reduce(engine.grad_flat)
            e1.record()
            torch.cuda.synchronize()
            allreduce_ms = e0.elapsed_time(e1) / 5
        except Exception as exc:                                       # never let the probe break the benchmark line
            log(f"all-reduce probe skipped: {exc}")

    if rank == 0:
        rays_total = world * n * args.steps
        points_per_launch = n * CHEST["n_samples"]
        sizes = {"bf16": (2, 2), "fp32": (4, 4)}[args.precision]
        kernels = {k: {"launches": c, "total_ms": ms} for k, (c, ms) in prof.items()}
        groups = {"hash_forward": ["encode_kernel"], "hash_backward": ["hash_backward_kernel", "scatter_bin_kernel", "scatter_reduce_kernel"],
                  "mlp_forward": ["mlp_forward_kernel"], "mlp_backward": ["mlp_backward_kernel", "mlp_grad_reduce_kernel"],
                  "adam": ["adam_kernel"]}
        ms_per_step = {g: sum(v["total_ms"] for k, v in kernels.items() if any(k.startswith(m) for m in members)) / args.steps
                       for g, members in groups.items()}
        ms_per_step = {g: v for g, v in ms_per_step.items() if v > 0}
        mfma_peak = 2500.0 if args.precision == "bf16" else 157.3      # TFLOP/s dense, MI355X_MICROARCH.md

        def roof(group):
            t = ms_per_step[group] * 1e-3
            members = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in kernels.items()
                       if any(k.startswith(m) for m in groups[group])}
            base = {"kernel": group, "per_step_ms": round(ms_per_step[group], 4), "points_per_step": points_per_launch,
                    "member_kernels_avg_launch_ms": members}       # HIP-event average per launch; compare with the rocprofv3 CSV
            if group in ("hash_forward", "hash_backward"):
                bytes_pp = algorithmic_bytes_per_point("encode_kernel" if group == "hash_forward" else "hash_backward_kernel", *sizes)
                achieved = bytes_pp * points_per_launch / t / 1e9
                traffic = None
                tf = os.path.join(REPO, "profiles", "pmc_traffic.json")      # filled in from rocprofv3 --pmc passes
                if os.path.exists(tf):
                    traffic = json.load(open(tf)).get(args.precision, {}).get(group)
                base.update({"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_point": bytes_pp})
            else:
                flops_pp = {"mlp_forward": 8256, "mlp_backward": 3 * 8256}[group]
                achieved = flops_pp * points_per_launch / t / 1e12
                base.update({"bound": "mfma", "achieved": round(achieved, 2), "peak": mfma_peak, "unit": "TFLOP/s",
                             "frac": round(achieved / mfma_peak, 4), "traffic": None, "algorithmic_flops_per_point": flops_pp})
            return base

        dominant = max((g for g in ms_per_step if g != "adam"), key=lambda g: ms_per_step[g])
        roofline = roof(dominant)
        out = {
            "metric": "train rays/sec, chest 256^3 / 50 proj", "value": rays_total / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"chest_50.yaml: 256^3 volume, 50 cone-beam projections 512x512, hash L=16 T=2^19 C=2 "
                                   f"({args.precision} table), S=192, MLP 32-32-32-(64)-32-1, Adam; "
                                   f"{n} rays/step/GPU (reference n_rays=1024), perturb=True",
                       "rays_per_step_per_gpu": n, "n_samples": CHEST["n_samples"], "parallelism": f"dp{world}"},
            "final_loss": final_loss, "scatter_overflow_last_step": overflow,
            "allreduce_ms_per_step": None if allreduce_ms is None else round(allreduce_ms, 4),
            "allreduce_bytes": engine.grad_flat.numel() * 4,
            "roofline": roofline,
            "roofline_hash_forward": roof("hash_forward"),
            "roofline_all": {g: roof(g) for g in ms_per_step if g != "adam"},
            "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in sorted(kernels.items())},
        }
        if world == 1 and args.cpu_seconds > 0:
            log("cpu_baseline leg (oracle, host cores)")
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, 1024, args.seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
