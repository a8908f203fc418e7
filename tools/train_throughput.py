#!/usr/bin/env python3
"""End-to-end rays/s of `python train.py --config config/chest_50.yaml`'s training loop (Trainer.start: dataset item ->
train_step -> StepLR, reference src/trainer.py:84-131) against the engine-only step on the same batch size.

    python tools/train_throughput.py [--epochs 6] [--table bfloat16]

The scan is the synthetic chest_50 (256^3 phantom, 50 cone-beam projections of 512x512) built in memory with the pickle
schema; evaluation and checkpoints are switched off so that only the training loop is timed."""
import argparse
import copy
import importlib.util
import json
import os
import sys
import tempfile
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from neuralvolumetricreconstructionformedicalimages_amd.config import load_config  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd.dataset import synthetic_scan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default=os.path.join(REPO, "config", "chest_50.yaml"))
ap.add_argument("--epochs", type=int, default=6, help="timed epochs of 50 steps each")
ap.add_argument("--table", default="bfloat16", choices=["float32", "bfloat16"])
ap.add_argument("--n-voxel", type=int, default=256)
args = ap.parse_args()

spec = importlib.util.spec_from_file_location("naf_train_entry", os.path.join(REPO, "train.py"))
entry = importlib.util.module_from_spec(spec)
spec.loader.exec_module(entry)

cfg = load_config(args.config)
data = synthetic_scan(n_voxel=args.n_voxel, n_train=50, n_val=2, device="cuda")
with tempfile.TemporaryDirectory() as tmp:
    cfg = copy.deepcopy(cfg)
    cfg["exp"].update(expdir=tmp, datadir=data)
    cfg["log"].update(i_eval=0, i_save=0, i_log=0)
    cfg["train"]["epoch"] = 1                                   # warm-up: epochs 0 and 1
    cfg.setdefault("backend", {})["table_dtype"] = args.table
    trainer = entry.BasicTrainer(cfg, torch.device("cuda"))
    n_rays, steps_per_epoch = cfg["train"]["n_rays"], len(trainer.train_dset)
    trainer.start()
    torch.cuda.synchronize()
    trainer.epoch_start, trainer.epochs = 2, 1 + args.epochs
    t0 = time.perf_counter()
    trainer.start()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    steps = args.epochs * steps_per_epoch
    loop = {"steps": steps, "ms_per_step": el / steps * 1e3, "rays_per_s": steps * n_rays / el}

    # engine-only: the same engine, one resident batch, no dataset / trainer code in the loop
    item = trainer.train_dset[0]
    rays, projs = item["rays"], item["projs"]
    weight = trainer.ray_weights(item, rays.shape[0])
    for i in range(20):
        trainer.engine.train_step(rays, projs, weight, ray_base=i * n_rays)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        trainer.engine.train_step(rays, projs, weight, ray_base=i * n_rays)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    eng = {"steps": steps, "ms_per_step": el / steps * 1e3, "rays_per_s": steps * n_rays / el}
print(json.dumps({"config": os.path.basename(args.config), "n_rays": n_rays, "table": args.table, "train_py_loop": loop,
                  "engine_only": eng, "loop_over_engine": loop["rays_per_s"] / eng["rays_per_s"]}))
