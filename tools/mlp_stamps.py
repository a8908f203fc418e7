#!/usr/bin/env python3
"""Phase stamps of mlp16_backward_kernel (library built with -DNAF_MLP_STAMPS, tools/build_variant.sh): shader-clock cycles from kernel
entry to the end of the weight-fragment build, of the tile loop, of the barrier behind it, of the four-wave fold and of the slab store,
per workgroup; plus the 100 MHz wall clock at entry and exit (launch skew between workgroups).
    python tools/mlp_stamps.py [--rays 1024]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neuralvolumetricreconstructionformedicalimages_amd import fused  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=1024)
args = ap.parse_args()
dev = torch.device("cuda")
scan = bench.ChestScan(dev, 1234, with_volume=False)
eng = bench.make_chest_engine(dev, "bf16", None, None, 0)
n = args.rays
rays = torch.empty(n, 8, device=dev)
w, _ = bench.step_weights(n, dev)
for i in range(5):
    tgt, _ = scan.sampler.draw(i, n, rays)
    eng.train_step(rays, tgt, w, ray_base=i * n)
torch.cuda.synchronize()
cfg = eng._cfg(0)
ws = fused.workspace(cfg, n * eng.n_samples, eng.device)        # the grow-only buffer is keyed by the device object
feat_bytes = (n * eng.n_samples * 32 * 2 + 255) & ~255
raw = ws.view(torch.uint8)[2 * feat_bytes:2 * feat_bytes + 768 * 4352 * 4].view(torch.int32).reshape(768, 4352)[:, 4300:4308].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
if not (raw[:, 5] != 0).any():                                     # not a stamps build, or another slab count: say what is there
    region = ws.view(torch.uint8)[2 * feat_bytes:2 * feat_bytes + 1024 * 4352 * 4].view(torch.int32).reshape(1024, 4352)
    print("no stamps; non-zero words per slab column range 4290..4320 of slab 0/1:", region[0, 4290:4320].tolist(), region[1, 4290:4320].tolist(),
          "non-zero slabs:", int((region[:, :64] != 0).any(dim=1).sum()), file=sys.stderr)
names = ["entry", "fragments built", "tile loop done", "barrier", "fold done", "slab stored"]
live = raw[:, 5] != 0
r = raw[live]
print(json.dumps({"rays": n, "workgroups_with_stamps": int(live.sum()),
                  "median_cycles_since_entry": {names[i]: float(np.median(r[:, i])) for i in range(6)},
                  "p90_cycles_since_entry": {names[i]: float(np.percentile(r[:, i], 90)) for i in range(6)},
                  "entry_wall_clock_spread_us": float(((r[:, 6] - r[:, 6].min()) & 0xFFFFFFFF).max() / 100.0),
                  "kernel_wall_us_first_entry_to_last_exit": float(((r[:, 7] - r[:, 6].min()) & 0xFFFFFFFF).max() / 100.0),
                  "median_workgroup_wall_us": float(np.median((r[:, 7] - r[:, 6]) & 0xFFFFFFFF) / 100.0)}))
