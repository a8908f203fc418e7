#!/usr/bin/env python3
"""Time-to-PSNR of the chest_50 reconstruction for a grid of operating points (rays per step x learning rate):

    python tools/psnr_race.py --configs 1024:1e-3,4096:1e-3,4096:2e-3,16384:4e-3 --max-train-s 10 --out gpurun_out/race.jsonl

Each run trains the synthetic chest scan from the same initial weights with bench.py's own step (device-side pixel draw, fused
forward / backward / Adam) and reports the TRAINING seconds, rays and steps to 30 / 35 / 38 dB volume PSNR (bench.psnr_race).
The reference's operating point is 1024 rays per step at lr 1e-3 (config/chest_50.yaml:29-30); everything else is an extension.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1024:1e-3", help="comma-separated rays:lr pairs")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--max-train-s", type=float, default=10.0)
    ap.add_argument("--burst-s", type=float, default=0.1)
    ap.add_argument("--thresholds", default="30,35,38")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    scan = bench.ChestScan(dev, 1234)
    ths = tuple(float(v) for v in args.thresholds.split(","))
    out = open(args.out, "a") if args.out else None
    for item in args.configs.split(","):
        n, lr = item.split(":")
        rec = bench.psnr_race(scan, int(n), float(lr), args.precision, ths, args.max_train_s, args.burst_s)
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()


if __name__ == "__main__":
    main()
