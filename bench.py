#!/usr/bin/env python3
"""bench.py -- training rays/s of the NAF hot path on chest_50 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = what `Trainer.train_step` of the reference does for one batch (src/trainer.py:134-142 around train.py:48-135,
data side src/dataset/tigre.py:354-372), all of it inside the timed region and none of it touching the host:
    draw distinct valid pixels of the step's projections on the device -> gather their measured values ->
    cone-beam ray generation -> fused render forward -> masked MSE -> backward (MLP + hash-table scatter) ->
    bucketed gradient all-reduce overlapped with the scatter (N > 1) -> Adam over the table and the MLP.
Workload: chest_50.yaml -- 256^3 volume, 50 cone-beam projections of 512x512, L=16 T=2^19 C=2 hash grid stored in bf16,
S=192 samples per ray, synthetic phantom with analytic projections (the reference ships no data).  The scan (poses,
projections, per-projection valid-pixel lists) is resident in HBM before the clock starts.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CHEST = dict(n_voxel=256, n_proj=50, n_samples=192, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
             bound=0.3, lr=1e-3, yaml_rays=1024)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RAYS_PER_PROJECTION = 16384    # a step of n rays draws from n / 16384 consecutive projections (the YAML step: 1024 from one)
# PSNR races run with every N = 1 bench (time / rays / steps to 30, 35, 38 dB volume PSNR from scratch): the reference's operating
# point (config/chest_50.yaml:29-30: 1024 rays, lr 1e-3) as the control, the same step at the learning rate that wins the grid of
# profiles/round3_psnr_race_grid.jsonl (an extension: the reference never changes lr), and the large batch at ITS best rate.
RACES = {"yaml_step_reference_lr": (1024, 1e-3), "yaml_step_lr_4e-3": (1024, 4e-3), "large_batch_16384_lr_4e-3": (16384, 4e-3)}


def algorithmic_bytes_per_point(kernel, table_bytes, feat_bytes, L=16, C=2, D=3):
    """hash fwd = D*4 + L*2^D*C*s_t + L*C*s_o ; hash bwd = L*C*s_g + D*4 + L*2^D*C*4*2 (fp32 atomic RMW).
    In the fused pipeline the sample position is recomputed from the 32-byte ray record, so the D*4 term is
    replaced by 32 B per ray / S samples (negligible); we keep the SURVEY formula for comparability."""
    if kernel == "encode_kernel":
        return D * 4 + L * (2 ** D) * C * table_bytes + L * C * feat_bytes
    if kernel == "hash_backward_kernel":
        return L * C * feat_bytes + D * 4 + L * (2 ** D) * C * 4 * 2
    raise KeyError(kernel)


def cpu_baseline(seconds, seed=0):
    """The oracle's pure-PyTorch CPU training step (the reference has no CPU hash encoder: SURVEY.md 8c/8d), chest_50
    shapes, fp32, the job's host threads.  Protocol of BASELINE.md section 3 -- 3 warm-up + 10 timed optimiser steps,
    median, at 1 024 and 8 192 rays per step -- cut short where the time budget runs out (the report says how many steps
    were timed): the 1 024-ray leg may use 60 % of `seconds`, the 8 192-ray leg the rest (at least one timed step each)."""
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

    def batch(n):
        ang = torch.rand(n, generator=g) * 3.1416
        o = torch.stack([torch.cos(ang), torch.sin(ang), torch.zeros(n)], -1)
        d = (torch.rand(n, 3, generator=g) - 0.5) * 0.25 - o
        rays = torch.cat([o, d, torch.full((n, 1), 0.814), torch.full((n, 1), 1.186)], -1)
        return rays, torch.rand(n, generator=g) * 0.1

    def step(rays, target):
        opt.zero_grad()
        acc = R.render(rays, net, None, CHEST["n_samples"], 0, True, 409600, 0.0)["acc"]
        ((acc - target) ** 2).mean().backward()
        opt.step()

    def leg(n, budget, warmups, want, est=None):
        """Up to `warmups` untimed and `want` timed steps; whenever the remaining budget cannot hold the remaining plan,
        warm-up steps are dropped first, then timed ones (never below one timed step)."""
        rays, target = batch(n)
        t_leg = time.perf_counter()
        step(rays[:64], target[:64])                     # pages the table in, builds autograd caches
        warm, times = 0, []
        while len(times) < want:
            left = budget - (time.perf_counter() - t_leg)
            timing = warm >= warmups or (est is not None and left < (warmups - warm + 2) * est)
            if timing and times and left < est:
                break
            t0 = time.perf_counter()
            step(rays, target)
            est = time.perf_counter() - t0
            if timing:
                times.append(est)
            else:
                warm += 1
        med = statistics.median(times)
        return {"rays_per_step": n, "warmup_steps": warm, "timed_steps": len(times), "median_step_s": round(med, 4),
                "rays_per_s": n / med}

    t0 = time.perf_counter()
    small = leg(1024, 0.6 * seconds, 3, 10)
    big = leg(8192, max(seconds - (time.perf_counter() - t0), 1.0), 1, 10, est=8 * small["median_step_s"])
    el = time.perf_counter() - t0
    return {"value": small["rays_per_s"], "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"median of {small['timed_steps']} optimiser steps of 1024 rays x {CHEST['n_samples']} samples after "
                      f"{small['warmup_steps']} warm-up steps (chest_50 shapes, fp32, oracle/ pure-PyTorch path, torch "
                      f"{torch.__version__}); 8192-ray leg: median of {big['timed_steps']} steps; {el:.0f} s of CPU work in all",
            "legs": [small, big]}


class ScanSampler:
    """The data side of a step on the device (tigre.py:354-372): distinct valid pixels of the step's projections, their
    measured values and their rays from ONE launch of `naf_draw_scan_rays`, no host synchronisation.  Valid-pixel lists
    are found once per projection (the reference recomputes `projs[index] > 0` for every item)."""

    def __init__(self, raygen, projs, seed):
        self.raygen, self.projs, self.seed = raygen, projs, int(seed)       # projs: [n_proj * H * W] fp32, resident
        hw = raygen.pixels_per_projection
        self.n_proj = projs.numel() // hw
        self.valid = [(torch.nonzero(projs[i * hw:(i + 1) * hw].abs() > 0, as_tuple=False).reshape(-1) + i * hw).contiguous()
                      for i in range(self.n_proj)]

    def draw(self, step, n, rays_out, target_out=None):
        per = min(n, RAYS_PER_PROJECTION)
        k = (n + per - 1) // per
        if k * per != n:
            raise ValueError(f"--rays must be <= {RAYS_PER_PROJECTION} or a multiple of it")
        lists = [self.valid[(step * k + j) % self.n_proj] for j in range(k)]
        seed = (self.seed * 0x9E3779B97F4A7C15 + (step + 1) * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
        targets = []
        for g0 in range(0, k, 16):                          # one launch per 16 projections (65 536 rays: one launch)
            part = lists[g0:g0 + 16]
            _, t, _ = self.raygen.draw(part, per, (seed + g0) & (2 ** 64 - 1), projections=self.projs,
                                       rays_out=rays_out[g0 * per:(g0 + len(part)) * per], want_pixels=False,
                                       target_out=None if target_out is None else target_out[g0 * per:(g0 + len(part)) * per])
            targets.append(t)
        return (targets[0] if len(targets) == 1 else torch.cat(targets)), rays_out

    def plan(self, step, n, rays_out, target_out):
        """The draw of `step` as a plan (struct naf_next_draw) for `NAFEngine.train_step(..., next_draw=...)`: the previous step takes
        it along.  None when the draw needs more than one launch (more than 16 projections per step)."""
        per = min(n, RAYS_PER_PROJECTION)
        k = (n + per - 1) // per
        if k * per != n or k > 16:
            return None
        lists = [self.valid[(step * k + j) % self.n_proj] for j in range(k)]
        seed = (self.seed * 0x9E3779B97F4A7C15 + (step + 1) * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
        return self.raygen.plan_draw(lists, per, seed, self.projs, rays_out, target_out)

    def draw_ranks(self, step, n, world, rays_out, target_out=None):
        """The draws of all `world` ranks of a step in ONE launch, in rank order (rank r: n distinct valid pixels of projection
        (step * world + r) mod n_proj): a level-parallel rank needs every rank's rays, and a shared seed makes them local."""
        if n > RAYS_PER_PROJECTION or world > 16:
            raise ValueError("draw_ranks: at most 16 ranks of at most one projection's draw each")
        lists = [self.valid[(step * world + r) % self.n_proj] for r in range(world)]
        seed = (self.seed * 0x9E3779B97F4A7C15 + (step + 1) * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
        _, t, _ = self.raygen.draw(lists, n, seed, projections=self.projs, rays_out=rays_out, want_pixels=False, target_out=target_out)
        return t, rays_out


class StepFeed:
    """Double-buffered rays / targets of consecutive steps: step k's launches carry the pixel draw of step k + 1 (the data side of
    tigre.py:354-372 stays inside the step -- one draw per step -- it just does not wait in line any more)."""

    def __init__(self, sampler, n_rays, device):
        self.sampler, self.n = sampler, n_rays
        self.rays = [torch.empty(n_rays, 8, device=device) for _ in range(2)]
        self.target = [torch.empty(n_rays, device=device) for _ in range(2)]
        self.ready = None                                              # the step whose rays sit in buffer (step & 1)

    def step(self, engine, i, weight, ray_base):
        cur = i & 1
        if self.ready != i:                                            # first step of a loop: draw it now
            self.sampler.draw(i, self.n, self.rays[cur], self.target[cur])
        nxt = self.sampler.plan(i + 1, self.n, self.rays[cur ^ 1], self.target[cur ^ 1])
        loss = engine.train_step(self.rays[cur], self.target[cur], weight, ray_base=ray_base, next_draw=nxt)
        self.ready = i + 1 if nxt is not None else None
        return loss


class ChestScan:
    """chest_50 synthetic scan resident in HBM: geometry, poses, all 50 x 512 x 512 measured values, the per-projection
    valid-pixel lists, and (for the PSNR half of the metric) the 256^3 ground-truth volume with its voxel-grid axes."""

    def __init__(self, device, sampler_seed, with_volume=True):
        from neuralvolumetricreconstructionformedicalimages_amd import phantom
        from neuralvolumetricreconstructionformedicalimages_amd.geometry import ConeGeometry, RayGenerator
        self.device = device
        self.geo = geo = ConeGeometry(phantom.scan_geometry(CHEST["n_voxel"], "cone"))
        angles = np.linspace(0, np.pi, CHEST["n_proj"] + 1)[:-1]          # generateData.py:175, totalAngle 180
        self.raygen = raygen = RayGenerator(geo, angles, device)
        table = phantom.ellipsoid_table(seed=0, extent=float(geo.sVoxel[0]) / 2)
        hw = raygen.pixels_per_projection
        self.projs = torch.empty(raygen.n_projections * hw, device=device)
        for i in range(raygen.n_projections):
            r = raygen.rays_for_projection(i)
            for j in range(0, hw, 1 << 16):
                self.projs[i * hw + j:i * hw + j + (1 << 16)] = phantom.line_integrals(r[j:j + (1 << 16)], table)
        self.sampler = ScanSampler(raygen, self.projs, sampler_seed)
        self.image = phantom.volume(geo, table, device=device).double() if with_volume else None
        s = geo.sVoxel / 2 - geo.dVoxel / 2                               # voxel centres, tigre.py:388-400
        self.axes = ([-float(v) for v in s], [float(v) for v in s], [int(v) for v in geo.nVoxel])

    def volume_psnr(self, net):
        """Reconstructed-volume PSNR of train.py:246-270: the whole 256^3 grid queried through `naf_field_forward_grid`,
        scored like `get_psnr_3d` (util.py:55-84: float64, 20 log10(1 / rmse), PIXEL_MAX = 1) -- evaluated on the device."""
        from neuralvolumetricreconstructionformedicalimages_amd.fused import field_query_grid
        vol = field_query_grid(net, *self.axes).double()
        mse = float((vol - self.image).square().mean().item())
        return 100.0 if mse <= 0.0 else 20.0 * float(np.log10(1.0 / np.sqrt(mse)))


def make_chest_engine(device, precision="bf16", lr=None, group=None, seed=0, **kw):
    from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    from neuralvolumetricreconstructionformedicalimages_amd.network import DensityNetwork
    torch.manual_seed(seed)                                           # identical initial weights on every rank
    enc = HashEncoder(3, CHEST["num_levels"], CHEST["level_dim"], CHEST["base_resolution"], CHEST["log2_hashmap_size"])
    net = DensityNetwork(enc, bound=CHEST["bound"], num_layers=4, hidden_dim=32, skips=[2], out_dim=1,
                         last_activation="sigmoid").to(device)
    tdt = torch.bfloat16 if precision == "bf16" else torch.float32
    return NAFEngine(net, CHEST["n_samples"], perturb=True, lr=CHEST["lr"] if lr is None else lr, table_dtype=tdt, seed=seed,
                     process_group=group, **kw)


def step_weights(n, device, world=1):
    """Per-ray loss weights of a step of n rays: the reference's sum of 200-ray chunk means at its own batch size
    (train.py:69,127), the plain (global) mean for every other batch."""
    from neuralvolumetricreconstructionformedicalimages_amd.loss import chunk_mean_weights
    if n == CHEST["yaml_rays"] and world == 1:
        return chunk_mean_weights(torch.ones(n, dtype=torch.bool, device=device), 200, "chunk_sum"), "chunk_sum (sum of 200-ray chunk means)"
    return torch.full((n,), 1.0 / (n * world), device=device), "global mean"


def psnr_race(scan, n_rays, lr, precision="bf16", thresholds=(30.0, 35.0, 38.0), max_train_s=12.0, burst_s=0.1, seed=0):
    """The PSNR half of the metric as a race: train chest_50 from scratch with steps of `n_rays` rays at learning rate `lr`
    and record the TRAINING time (seconds inside train steps: pixel draw + ray generation + forward + backward + Adam, evaluation
    excluded), the rays and the steps after which the reconstructed 256^3 volume first reaches each PSNR threshold.  The volume
    is evaluated between bursts of about `burst_s` seconds of training, so a reported time is an upper bound by at most one
    burst.  Also returns the sustained throughput over the whole race (rays / training seconds, >= 2 s unless the last
    threshold falls earlier)."""
    device = scan.device
    rays = torch.empty(n_rays, 8, device=device)
    weight, loss_name = step_weights(n_rays, device)

    feed = StepFeed(scan.sampler, n_rays, device)

    def run(engine, first, count):
        for i in range(first, first + count):
            feed.step(engine, i, weight, i * n_rays)

    scratch = make_chest_engine(device, precision, lr, seed=seed)      # sizes the workspace, measures the step for the burst length
    run(scratch, 0, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(scratch, 3, 10)
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 10
    del scratch
    burst = max(1, int(round(burst_s / est)))
    engine = make_chest_engine(device, precision, lr, seed=seed)
    curve = [{"train_s": 0.0, "steps": 0, "psnr_db": round(scan.volume_psnr(engine.net), 3)}]
    reached, pending = {}, sorted(thresholds)
    t_train, steps = 0.0, 0
    while pending and t_train < max_train_s:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(engine, steps, burst)
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t0
        steps += burst
        psnr = scan.volume_psnr(engine.net)
        curve.append({"train_s": round(t_train, 4), "steps": steps, "psnr_db": round(psnr, 3)})
        while pending and psnr >= pending[0]:
            reached[f"{pending.pop(0):g}dB"] = {"train_s": round(t_train, 4), "steps": steps, "rays": steps * n_rays}
    for th in pending:
        reached[f"{th:g}dB"] = None                                     # not reached inside max_train_s
    # The curve is not monotone (at the reference's learning rate it oscillates by +-1.5 dB between 32 and 35 dB for seconds), so the
    # FIRST crossing of a threshold can fall on a lucky peak: `held_from` is the training time from which the PSNR stayed at or above
    # the threshold until the end of the race (for the highest threshold reached that is its first crossing: the race stops there).
    held = {}
    for th in sorted(thresholds):
        below = [i for i, c in enumerate(curve) if c["psnr_db"] < th]
        first_ok = (below[-1] + 1) if below else 0
        held[f"{th:g}dB"] = None if first_ok >= len(curve) else {"train_s": curve[first_ok]["train_s"], "steps": curve[first_ok]["steps"]}
    keep = curve[::max(1, len(curve) // 24)]
    if keep[-1] is not curve[-1]:
        keep.append(curve[-1])
    # `time_to_psnr` is the HELD figure (a first crossing is a lucky peak of a curve that oscillates by +-1.5 dB);
    # the first crossings stay in the record under their own name.
    return {"rays_per_step": n_rays, "lr": lr, "precision": precision, "loss": loss_name, "burst_steps": burst,
            "time_to_psnr": held, "first_crossing": reached, "final": curve[-1], "train_seconds": round(t_train, 4), "steps": steps,
            "sustained_rays_per_s": steps * n_rays / t_train, "ms_per_step": round(t_train / steps * 1e3, 4), "curve": keep}


def launch_ranks(n_gpus):
    """`python bench.py --gpus N` with no torchrun environment: start `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a CHILD process (never exec: this process may not replace itself once anything has touched the GPU, and nothing here has),
    let rank 0's JSON line through on the inherited stdout and return the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {n_gpus} without WORLD_SIZE: launching {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, cwd=REPO)


def full_schedule(scan, precision, seed=0, epochs=None, eval_epochs=25, thresholds=(30.0, 35.0, 38.0), log=None):
    """The PSNR half of the metric on the reference's OWN schedule (config/chest_50.yaml:27-33 + src/trainer.py:54-58,83-132): `epoch`
    epochs of one 1 024-ray step per projection (50 x 1 500 = 75 000 steps, 76.8 M rays), Adam at `lrate` with StepLR(lrate_step,
    lrate_gamma) stepped once per epoch, loss = sum of 200-ray chunk means -- from scratch, in the given precision.  Pixel draws
    (sampler seed, step) and jitter (engine seed, ray index) do not depend on the precision, so a bf16 and an fp32 run see identical
    data.  Returns the final reconstructed-volume PSNR, the training seconds (evaluation excluded) and the PSNR curve."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REPO, "config", "chest_50.yaml")))["train"]
    n_rays, lr0, gamma, lr_step = int(cfg["n_rays"]), float(cfg["lrate"]), float(cfg["lrate_gamma"]), int(cfg["lrate_step"])
    epochs = int(cfg["epoch"]) if epochs is None else int(epochs)
    per_epoch = scan.raygen.n_projections                          # len(train_dloader): one item per projection, n_batch = 1
    device = scan.device
    rays = torch.empty(n_rays, 8, device=device)
    weight, loss_name = step_weights(n_rays, device)
    engine = make_chest_engine(device, precision, lr0, seed=seed)
    feed = StepFeed(scan.sampler, n_rays, device)
    curve = [{"train_s": 0.0, "epoch": 0, "steps": 0, "psnr_db": round(scan.volume_psnr(engine.net), 3)}]
    t_train, step = 0.0, 0
    for e0 in range(0, epochs, eval_epochs):
        e1 = min(epochs, e0 + eval_epochs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e in range(e0, e1):
            engine.lr = lr0 * gamma ** (e // lr_step)              # StepLR: the scheduler steps at the END of every epoch (trainer.py:130)
            for _ in range(per_epoch):
                feed.step(engine, step, weight, step * n_rays)
                step += 1
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t0
        curve.append({"train_s": round(t_train, 3), "epoch": e1, "steps": step, "psnr_db": round(scan.volume_psnr(engine.net), 3)})
        if log is not None and (e1 % 250 == 0 or e1 == epochs):
            log(f"  full schedule {precision}: epoch {e1}/{epochs}, {t_train:.1f} s of training, {curve[-1]['psnr_db']:.2f} dB")
    held = {}
    for th in thresholds:                                           # training time from which the PSNR stayed at or above the threshold
        below = [i for i, c in enumerate(curve) if c["psnr_db"] < th]
        k = (below[-1] + 1) if below else 0
        held[f"{th:g}dB"] = None if k >= len(curve) else {"train_s": curve[k]["train_s"], "steps": curve[k]["steps"]}
    tail = [c["psnr_db"] for c in curve[-8:]]
    return {"precision": precision, "epochs": epochs, "steps": step, "rays": step * n_rays, "rays_per_step": n_rays, "lr": lr0,
            "lr_schedule": f"StepLR(step_size={lr_step} epochs, gamma={gamma:g})", "loss": loss_name,
            "psnr_db": curve[-1]["psnr_db"], "psnr_db_last_8_evals_min_max": [min(tail), max(tail)],
            "train_seconds": round(t_train, 3), "rays_per_s": step * n_rays / t_train, "time_to_psnr": held,
            "curve": curve[::max(1, len(curve) // 30)] + ([curve[-1]] if (len(curve) - 1) % max(1, len(curve) // 30) else [])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.6 s of back-to-back steps at the reference's batch size (a 20-step region is 6 ms: clocks and caches have not settled)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--rays", type=int, default=CHEST["yaml_rays"],
                    help="rays per GPU per step (default: chest_50.yaml's n_rays = 1024, the step that reconstructs fastest)")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16", help="table storage / MLP operand type")
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="budget of the cpu_baseline leg (0 disables it)")
    ap.add_argument("--sub-records", type=int, default=1, help="0: skip the fp32-parity-mode and 1024-ray sub-records (N = 1 only)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lr", type=float, default=None, help="Adam learning rate (default: chest_50.yaml's 1e-3)")
    ap.add_argument("--psnr-seconds", type=float, default=12.0,
                    help="training-time budget of each PSNR race (time to 30 / 35 / 38 dB volume PSNR; N = 1 only; 0 disables)")
    ap.add_argument("--full-schedule", type=int, default=1,
                    help="1: train chest_50.yaml's own schedule (1 500 epochs x 50 steps of 1 024 rays) from scratch in bf16 AND in the fp32 "
                         "parity mode on identical draws and report both final volume PSNRs (N = 1 only; ~60 s); 0 disables")
    ap.add_argument("--full-schedule-epochs", type=int, default=None, help="diagnostics: shorten the full schedule")
    ap.add_argument("--scatter-mode", choices=["auto", "atomic", "binned"], default="auto",
                    help="table-gradient scatter: auto = binned from 2^13 points per step on (naf_render_cfg.scatter_mode)")
    ap.add_argument("--cfg-flags", type=int, default=0, help="diagnostics: extra naf_render_cfg.flags bits (e.g. 2048 = NAF_CFG_SCATTER_PAIR12)")
    ap.add_argument("--per-level", action="store_true", help="diagnostics: one launch per level (NAF_CFG_PER_LEVEL_LAUNCHES)")
    ap.add_argument("--interleaved-levels", action="store_true",
                    help="diagnostics: the encoder walks all levels of a point tile at once (NAF_CFG_LEVELS_INTERLEAVED)")
    ap.add_argument("--two-gathers", action="store_true",
                    help="diagnostics: the encoder fetches x-neighbour corners with two gathers instead of one 16-byte window "
                         "(NAF_CFG_ENCODE_TWO_GATHERS)")
    ap.add_argument("--windows", action="store_true", help="diagnostics: NAF_CFG_ENCODE_WINDOWS (16-byte window gathers at every batch size)")
    ap.add_argument("--bwd-one-wave", action="store_true", help="diagnostics: NAF_CFG_BACKWARD_ONE_WAVE_PER_SIMD")
    ap.add_argument("--encode-groups", type=int, choices=[0, 1, 2, 4, 8], default=0, help="XCD groups of the encoder (NAF_CFG_ENCODE_GROUPS_*; 1: level-major, 0: by batch size)")
    ap.add_argument("--separate-adam", action="store_true",
                    help="diagnostics: write the table gradient out and run the table's Adam pass as its own launch "
                         "(default: the gradient reducer applies it, naf_render_train_adam)")
    ap.add_argument("--force-dp", action="store_true",
                    help="diagnostics on one GPU: run the data-parallel step (bucketed scatter, RCCL collectives on the side "
                         "stream, per-bucket Adam) with a world-size-1 process group")
    ap.add_argument("--dp-mode", choices=["auto", "levels", "sharded", "allreduce"], default="sharded",
                    help="N > 1: levels = each rank owns L/N levels, two all-to-alls of features / feature gradients per step; sharded = "
                         "reduce-scatter of the table gradient -> per-rank Adam on a table slice -> all-gather; allreduce = all-reduce + "
                         "replicated Adam; auto = whichever puts fewer bytes on the links (levels below ~3 300 rays per GPU).  Default sharded: the "
                         "level-parallel step stays opt-in until it has run over RCCL on more than one rank (DESIGN.md section 6)")
    ap.add_argument("--buckets", default=None, help="level buckets of the data-parallel exchange, e.g. 8-16,0-8 (default: dist.default_bucket_levels -- one range below 2^20 points per step)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the ray batch is pipelined over (engine n_streams)")
    ap.add_argument("--chunk-rays", type=int, default=16384, help="rays per pipelined chunk when --streams > 1")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:                # bare `python bench.py --gpus N`: become the launcher
        raise SystemExit(launch_ranks(args.gpus))
    # ONE JSON line on stdout: libraries that print to file descriptor 1 (RCCL's version banner at the first collective) are
    # sent to stderr for the whole run; the result line is written to the real stdout at the end.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
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
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29555")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD

    from neuralvolumetricreconstructionformedicalimages_amd import _abi
    from neuralvolumetricreconstructionformedicalimages_amd.build import source_fingerprint

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    # ---- chest_50 scan, resident in HBM: geometry, poses, all 50 x 512 x 512 measured values -----------------------
    scan = ChestScan(device, 1234 + rank, with_volume=(rank == 0))    # each rank draws its own shard
    sampler = scan.sampler
    log(f"scan resident: {scan.raygen.n_projections} projections, {min(v.numel() for v in sampler.valid)}.."
        f"{max(v.numel() for v in sampler.valid)} valid pixels each")

    def make_engine(precision, group):
        buckets = None
        if args.buckets and group is not None:
            buckets = [tuple(int(v) for v in b.split("-")) for b in args.buckets.split(",")]
        return make_chest_engine(device, precision, args.lr, group, args.seed,
                                 n_streams=args.streams if group is None else 1, chunk_rays=args.chunk_rays,
                                 scatter_mode={"auto": 0, "atomic": 1, "binned": 2}[args.scatter_mode],
                                 cfg_flags=args.cfg_flags | (_abi.CFG_PER_LEVEL_LAUNCHES if args.per_level else 0) | (_abi.CFG_LEVELS_INTERLEAVED if args.interleaved_levels else 0)
                                           | (_abi.CFG_ENCODE_TWO_GATHERS if args.two_gathers else 0) | (_abi.CFG_ENCODE_WINDOWS if args.windows else 0) | (_abi.CFG_BACKWARD_ONE_WAVE_PER_SIMD if args.bwd_one_wave else 0) | {0: 0, 1: _abi.CFG_ENCODE_LEVEL_MAJOR, 2: _abi.CFG_ENCODE_GROUPS_2, 4: _abi.CFG_ENCODE_GROUPS_4, 8: _abi.CFG_ENCODE_GROUPS_2 | _abi.CFG_ENCODE_GROUPS_4}[args.encode_groups],
                                 bucket_levels=buckets, fuse_table_adam=not args.separate_adam, dp_mode=args.dp_mode,
                                 rays_per_step_hint=args.rays)

    engine = make_engine(args.precision, pg)
    n = args.rays
    levels_mode = engine.dp_mode == "levels" and pg is not None
    if levels_mode:
        fb = 4 if args.precision == "fp32" else 2
        allreduce_bytes = 2 * n * CHEST["n_samples"] * 32 * fb * (world - 1) // world + 4 * (engine.grad_flat.numel() - engine._emb_flat.numel())
    else:
        allreduce_bytes = engine.grad_flat.numel() * 4
    dp_buckets = None if engine._dp is None else [list(b) for b in engine._dp["levels"]]
    total_steps = args.warmup + args.steps
    rays = torch.empty(n, 8, device=device)
    weight, loss_name = step_weights(n, device, world)               # N > 1: global mean over all ranks' rays
    shared = None
    if levels_mode and n <= RAYS_PER_PROJECTION and world <= 16:     # every rank draws the step's whole pixel set: one seed for all
        shared = ScanSampler.__new__(ScanSampler)
        shared.raygen, shared.projs, shared.seed, shared.n_proj, shared.valid = sampler.raygen, sampler.projs, 1234, sampler.n_proj, sampler.valid
        rays_all = torch.empty(world * n, 8, device=device)
        target_all = torch.empty(world * n, device=device)

    feed = StepFeed(sampler, n, device) if (world == 1 and not args.force_dp) else None

    def step(i, eng=None, n_rays=n, ray_buf=rays, w=weight):
        # G6 + G3 in one launch: distinct valid pixels, their measured values, their cone-beam rays (no host round trip).
        # (Drawing step k + 1 on a side stream while step k computes was measured and is not done: the event waits that order the two
        # streams cost more stream time than the 8 us launch they hide -- 0.3182 against 0.3118 ms per step.)
        if shared is not None and eng is None:
            shared.draw_ranks(i, n_rays, world, rays_all, target_all)
            lo, hi = rank * n_rays, (rank + 1) * n_rays
            return engine.train_step(rays_all[lo:hi], target_all[lo:hi], w, ray_base=(i * world + rank) * n_rays, rays_all=rays_all)
        if feed is not None and eng is None and n_rays == n:           # the main loop: step i carries the draw of step i + 1
            return feed.step(engine, i, w, i * n_rays)
        target, _ = sampler.draw(i, n_rays, ray_buf)
        return (eng or engine).train_step(ray_buf, target, w, ray_base=(i * world + rank) * n_rays)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    log(f"starting {args.warmup} warm-up steps of {n} rays")
    for i in range(args.warmup):
        step(i)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    barrier()
    log(f"timing {args.steps} steps")
    # ---- the timed region: exactly K steps, nothing but the step inside (no event pairs around kernels: at 0.3 ms per step they
    # would be a tenth of what is measured) ------------------------------------------------------------------------------
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        loss = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed region {elapsed:.4f} s")
    final_loss = float(loss.item())
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel times: a second pass over the same kind of steps with a HIP-event pair around every kernel, on its launch
    # stream (naf_profile_*).  The events cost ~10 % at this step size, which is why this pass is not the timed region. ----------
    prof_steps = min(max(args.steps, 50 if n <= 4096 else 10), 200 if n <= 4096 else 20)
    engine.comm_timing(True)
    _abi.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(total_steps, total_steps + prof_steps):
        step(i)
    barrier()
    prof_elapsed = time.perf_counter() - t0
    prof = _abi.profile_collect()
    _abi.profile_enable(False)
    comm = engine.comm_report()
    engine.comm_timing(False)
    n_scatter = n * world if levels_mode else n               # a level-parallel rank scatters every rank's points (its levels)
    overflow = engine.scatter_overflow(n_scatter)
    overflow_levels = engine.scatter_overflow_levels(n_scatter)     # read HERE: the sub-records below re-lay the workspace out
    if overflow_levels is not None and overflow is not None and sum(overflow_levels) != overflow:
        raise SystemExit(f"scatter overflow counters disagree: total {overflow}, per level {overflow_levels}")

    def timed(fn, steps, warm):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(warm, warm + steps):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps

    sustained = None
    if world == 1 and not args.force_dp:
        k = max(args.steps, int(2.2 / (elapsed / args.steps)))           # >= 2 s of back-to-back steps
        dt = timed(lambda i: step(total_steps + prof_steps + i), k, 0)
        sustained = {"steps": k, "seconds": round(dt * k, 3), "ms_per_step": round(dt * 1e3, 4), "rays_per_s": n / dt}
        log(f"sustained: {k} steps in {dt * k:.2f} s")

    sub_records = None
    if rank == 0 and world == 1 and args.sub_records and not args.force_dp:
        log("sub-records: the large batch, the fp32 parity mode")
        # (a) the throughput end of the batch curve (the headline of rounds 1-2): 65 536 rays from 4 projections per step
        sub_records = {}
        if n != 65536:
            m = 65536
            r1, (w1, _) = torch.empty(m, 8, device=device), step_weights(m, device)
            dt = timed(lambda i: step(i, engine, m, r1, w1), 20, 3)
            sub_records["large_batch_65536_rays"] = {"rays_per_step": m, "precision": args.precision, "ms_per_step": round(dt * 1e3, 4),
                                                     "rays_per_s": m / dt, "loss": "global mean"}
            del r1, w1
        # (b) the reference's arithmetic: fp32 table, fp32 MFMA (bit-for-bit an fmaf chain), fp32 scatter records
        if args.precision == "bf16":
            e32 = make_engine("fp32", None)
            for m, steps in ((CHEST["yaml_rays"], 300), (16384, 10)):
                r2, (w2, ln) = torch.empty(m, 8, device=device), step_weights(m, device)
                dt = timed(lambda i: step(i, e32, m, r2, w2), steps, 5)
                sub_records[f"fp32_parity_mode_{m}_rays"] = {"rays_per_step": m, "precision": "fp32", "ms_per_step": round(dt * 1e3, 4),
                                                             "rays_per_s": m / dt, "loss": ln}
            del e32
        torch.cuda.empty_cache()

    # ---- the PSNR half of the metric: time / rays / steps to 30, 35, 38 dB volume PSNR, from scratch, per operating point -------
    psnr = None
    if rank == 0 and world == 1 and args.psnr_seconds > 0 and not args.force_dp:
        psnr = {}
        for name, (m, lr) in RACES.items():
            log(f"PSNR race {name}: {m} rays/step, lr {lr}")
            psnr[name] = psnr_race(scan, m, lr, args.precision, max_train_s=args.psnr_seconds)
            log(f"  -> {psnr[name]['time_to_psnr']}")

    # ---- ... and the same half on the reference's own schedule, in both precisions ------------------------------------------------
    schedule = None
    if rank == 0 and world == 1 and args.full_schedule and not args.force_dp:
        schedule = {}
        for prec in ("bf16", "fp32"):
            log(f"full schedule ({prec}): chest_50.yaml's epochs x projections x 1024 rays, from scratch")
            schedule[prec] = full_schedule(scan, prec, seed=args.seed, epochs=args.full_schedule_epochs, log=log)
            torch.cuda.empty_cache()
        schedule["psnr_db_abs_difference"] = round(abs(schedule["bf16"]["psnr_db"] - schedule["fp32"]["psnr_db"]), 3)

    if rank == 0:
        rays_total = world * n * args.steps
        points_per_launch = n * CHEST["n_samples"]
        n_params = engine.emb.numel()
        tb, fb = {"bf16": (2, 2), "fp32": (4, 4)}[args.precision]
        kernels = {k: {"launches": c, "total_ms": ms, "avg_ms": ms / max(c, 1)} for k, (c, ms) in prof.items()}
        mfma_peak = 2500.0 if args.precision == "bf16" else 157.3      # TFLOP/s dense, MI355X_MICROARCH.md
        # pair record of the binned scatter: 8 bytes in bf16 mode (scatter_v2.h; 12 with the round-3 kernels, --cfg-flags 2048), 20 in fp32 mode
        rec_bytes = (12 if args.cfg_flags & 2048 else 8) if args.precision == "bf16" else 20
        fused_adam = engine.fuse_table_adam and engine._dp is None
        # Algorithmic bytes / flops of ONE launch of each kernel.  encode: SURVEY 8(d)'s hash-forward figure.
        # The gradient scatter is two kernels here, each priced on what it must move: bin reads the feature gradients and writes
        # one pair record per x-neighbour corner pair; reduce reads the records and -- with the Adam tail -- reads and writes
        # parameter, both moments (fp32) and writes the 16-bit shadow of every table element (otherwise read + write of the gradient).
        algo = {
            "encode_kernel": ("hbm", algorithmic_bytes_per_point("encode_kernel", tb, fb) * points_per_launch),
            "scatter_bin_kernel": ("hbm", (32 * fb + 16 * 4 * rec_bytes) * points_per_launch),
            "scatter_reduce_kernel": ("hbm", 16 * 4 * rec_bytes * points_per_launch
                                      + n_params * ((24 + (2 if args.precision == "bf16" else 0)) if fused_adam else 8)),
            "hash_backward_kernel": ("hbm", algorithmic_bytes_per_point("hash_backward_kernel", tb, fb) * points_per_launch),
            "mlp_forward_kernel": ("mfma", 8256 * points_per_launch),
            "mlp_backward_kernel": ("mfma", 3 * 8256 * points_per_launch),
            "adam_kernel": ("hbm", None),
        }

        # HBM bytes by PMC counters come from separate rocprofv3 --pmc passes (tools/collect_profiles.sh); they are printed only
        # while the kernel sources are the ones those passes ran on
        traffic_table, traffic_note = {}, "profiles/pmc_traffic.json absent"
        tf = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            doc = json.load(open(tf))
            if doc.get("csrc_fingerprint") != source_fingerprint():
                traffic_note = (f"stale: the PMC passes ran on kernel sources {doc.get('csrc_fingerprint')}, this build is "
                                f"{source_fingerprint()} -- rerun tools/collect_profiles.sh + tools/install_profiles.py")
            elif str(n) not in doc.get("by_rays", {}) or world != 1:
                traffic_note = f"the PMC passes were taken at {sorted(doc.get('by_rays', {}))} rays/step on one GPU"
            else:
                traffic_table = doc["by_rays"][str(n)].get(args.precision, {})
                traffic_note = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at commit {doc.get('collected_at_commit')}, "
                                f"kernel sources {doc.get('csrc_fingerprint')}, {n} rays/step")

        def roof(kernel):
            """Roofline of ONE kernel: algorithmic bytes (or flops) of a launch / its average launch duration (HIP events on the
            launch stream, profiled pass); `traffic` = HBM bytes per launch by PMC; `real_frac` = traffic / time / peak."""
            v = kernels[kernel]
            t = v["avg_ms"] * 1e-3
            bound, work = algo.get(kernel, ("hbm", None))
            base = {"kernel": kernel, "avg_launch_ms": round(v["avg_ms"], 5), "launches_per_step": round(v["launches"] / prof_steps, 2),
                    "points_per_launch": points_per_launch, "bound": bound}
            if work is None:
                return base
            if bound == "hbm":
                ach = work / t / 1e9
                traffic = traffic_table.get(kernel)
                base.update({"achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                             "traffic": traffic, "algorithmic_bytes_per_launch": work,
                             "real_frac": None if not traffic else round(traffic / t / 1e9 / HBM_PEAK_GBS, 4)})
            else:
                ach = work / t / 1e12
                base.update({"achieved": round(ach, 2), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(ach / mfma_peak, 4),
                             "traffic": None, "algorithmic_flops_per_launch": work})
            return base

        per_step = {k: v["total_ms"] / prof_steps for k, v in kernels.items()}
        dominant = max((k for k in per_step if k in algo and algo[k][1] is not None), key=lambda k: per_step[k])
        kernel_ms = sum(per_step.values())
        out = {
            "metric": "train rays/sec, chest 256^3 / 50 proj", "value": rays_total / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"chest_50.yaml: 256^3 volume, 50 cone-beam projections 512x512, hash L=16 T=2^19 C=2 "
                                   f"({args.precision} table), S=192, MLP 32-32-32-(64)-32-1, Adam lr {engine.lr:g}; "
                                   f"{n} rays/step/GPU (reference n_rays=1024) = {min(n, RAYS_PER_PROJECTION)} distinct valid pixels from "
                                   f"{'one projection' if n <= RAYS_PER_PROJECTION else f'each of {n // RAYS_PER_PROJECTION} projections'}, "
                                   f"drawn on the device inside the step; perturb=True; loss: {loss_name}"
                                   + ("; the operating point that reaches 35 dB volume PSNR soonest (psnr.*, profiles/round3_psnr_race_grid.jsonl)"
                                      if n == CHEST["yaml_rays"] else ""),
                       "rays_per_step_per_gpu": n, "n_samples": CHEST["n_samples"], "lr": engine.lr, "parallelism": (f"lp{world}" if levels_mode else f"dp{world}")},
            "final_loss": final_loss, "sustained": sustained, "psnr": psnr, "full_schedule": schedule,
            "scatter_overflow_last_step": overflow, "scatter_overflow_levels": overflow_levels,
            "library_kernels_ms_per_step": round(kernel_ms, 4),
            "profiled_pass": {"steps": prof_steps, "ms_per_step": round(prof_elapsed / prof_steps * 1e3, 4),
                              "note": "per-kernel HIP-event pairs switched on; not the timed region"},
            "allreduce_ms_per_step": None, "allreduce_exposed_ms_per_step": None, "allreduce_buckets": dp_buckets,
            "allreduce_bytes": allreduce_bytes, "rays_per_s_per_gpu": rays_total / elapsed / world,
            "grad_exchange": ("level-parallel: rank k owns levels [k L/N, (k+1) L/N) -- all-to-all of the features, all-to-all of the feature "
                              "gradients, Adam on the owned rows; MLP gradient + loss all-reduced (allreduce_bytes = bytes a rank sends per step)")
            if levels_mode else None if engine._dp is None else (
                "reduce-scatter of the table gradient -> Adam on this rank's 1/N slice -> all-gather of the table the kernels read; "
                "MLP gradient + loss all-reduced" if engine.dp_mode == "sharded" else "all-reduce per level bucket, Adam replicated"),
            "roofline": roof(dominant),
            "roofline_hash_forward": roof("encode_kernel") if "encode_kernel" in kernels else None,
            "roofline_all": {k: roof(k) for k in sorted(kernels) if k in algo},
            "traffic_source": traffic_note,
            "kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step.items())},
        }
        if comm is not None:
            # in flight: the collectives on the side stream, measured by events inside the profiled steps.  exposed: what the main
            # stream waited after its own compute = tail (end of compute -> last Adam launched) minus the Adam kernels themselves
            out["allreduce_ms_per_step"] = round(comm["allreduce_ms_per_step"], 4)
            if levels_mode:                              # both all-to-alls sit on the critical path; the phases of a step by HIP events
                out["allreduce_exposed_ms_per_step"] = round(comm["tail_ms_per_step"], 4)
                out["level_parallel_phases_ms"] = {k: round(v, 4) for k, v in comm.items() if k.endswith("_ms")}
            else:
                out["allreduce_exposed_ms_per_step"] = round(max(0.0, comm["tail_ms_per_step"] - per_step.get("adam_kernel", 0.0)), 4)
        if sub_records is not None:
            out["sub_records"] = sub_records
            par = sub_records.get(f"fp32_parity_mode_{CHEST['yaml_rays']}_rays")
            if par is not None:
                # the mode that meets BOTH parity bars of the north star (projection rel. L2 1e-4 and PSNR +-0.1 dB): the reference's own
                # arithmetic (fp32 table, fp32 MFMA = an fmaf chain, fp32 scatter records), same YAML step, same box, same run
                out["parity_mode"] = {"value": par["rays_per_s"], "unit": "rays/s", "ms_per_step": par["ms_per_step"], "dtype": "f32",
                                      "rays_per_step": par["rays_per_step"], "loss": par["loss"]}
        if world == 1 and args.cpu_seconds > 0 and not args.force_dp:
            log("cpu_baseline leg (oracle, host cores)")
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.seed)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
