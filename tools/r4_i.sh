#!/bin/bash
# does the bf16 kernels' hardware sigmoid / dot2c bias sum move the final PSNR of the reference's schedule?  four seeds per library build
L=neuralvolumetricreconstructionformedicalimages_amd/lib
mkdir -p gpurun_out/r4i
for v in cur exact; do
  cp $L/ab/$v.so $L/libnaf_hip.so || exit 1
  timeout -k 10 500 python tools/precision_grid.py --epochs 1500 --combos bf16:bf16 --seeds 0,1,2,3 --out gpurun_out/r4i/seeds_$v.jsonl > gpurun_out/r4i/$v.log 2>&1
  python -c "
import json
for l in open('gpurun_out/r4i/seeds_$v.jsonl'):
    d=json.loads(l); print('$v', d['seed'], d['psnr_fp32_master_fp32_eval'], d['psnr_every_100_epochs'][4::5])"
done
