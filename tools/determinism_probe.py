#!/usr/bin/env python3
"""Is a training run bit-reproducible?  Two engines from the same seed take the same steps (chest_50 YAML step); after every step
the fp32 master table, its moments and the MLP are compared bit for bit.  Reports the first step at which they differ and where.
    python tools/determinism_probe.py [--precision fp32] [--steps 300] [--rays 1024]"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp32")
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--rays", type=int, default=1024)
args = ap.parse_args()
dev = torch.device("cuda", 0)
scan = bench.ChestScan(dev, 1234, with_volume=False)
n = args.rays
rays = torch.empty(n, 8, device=dev)
weight, _ = bench.step_weights(n, dev)
engines = [bench.make_chest_engine(dev, args.precision, bench.CHEST["lr"], seed=0) for _ in range(2)]
out = {"precision": args.precision, "rays": n, "steps": args.steps, "first_difference": None}
for step in range(args.steps):
    target, _ = scan.sampler.draw(step, n, rays)
    t = target.clone()
    losses = []
    for e in engines:
        e.train_step(rays, t, weight, ray_base=step * n)
        losses.append(e.loss.clone())
    a, b = engines
    diffs = {"loss": not torch.equal(losses[0], losses[1]), "table": not torch.equal(a.emb, b.emb), "m": not torch.equal(a.emb_m, b.emb_m),
             "v": not torch.equal(a.emb_v, b.emb_v), "mlp": not torch.equal(a.mlp, b.mlp)}
    if any(diffs.values()):
        d = (a.emb - b.emb).abs()
        offs = [int(v) for v in a.offsets.tolist()]
        rows = torch.nonzero(d.amax(dim=1) > 0).reshape(-1)
        lv = sorted({max(i for i, o in enumerate(offs[:-1]) if o <= int(r)) for r in rows[:2000].tolist()})
        out["first_difference"] = {"step": step, "what": [k for k, v in diffs.items() if v], "table_rows_differing": int(rows.numel()),
                                   "levels_of_first_2000_rows": lv, "max_abs_table_diff": float(d.max()),
                                   "mlp_max_abs_diff": float((a.mlp - b.mlp).abs().max())}
        break
print(json.dumps(out))
