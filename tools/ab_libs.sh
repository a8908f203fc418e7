#!/bin/bash
# Same-box A/B of library builds: box-to-box spread (+-3 %) is larger than most single changes, so variants are compared on ONE box.
# Here (no GPU):   build variant -> cp neuralvolumetricreconstructionformedicalimages_amd/lib/libnaf_hip.so .../lib/ab/<name>.so  (per variant; lib/ab/ travels
#                  with the snapshot like every built .so and is git-ignored)
# On the box:      gpurun -- 'bash tools/ab_libs.sh base nt base nt'   -> gpurun_out/ab_libs.jsonl, one line per (variant, batch size)
# The box's copy of lib/libnaf_hip.so is overwritten (it is scratch); rebuild here afterwards.
L=neuralvolumetricreconstructionformedicalimages_amd/lib
OUT=gpurun_out/ab_libs.jsonl
RAYS=${RAYS:-"1024 16384 65536"}
for v in "$@"; do
  cp $L/ab/$v.so $L/libnaf_hip.so || exit 1
  for r in $RAYS; do
    timeout -k 10 100 python tools/step_bench.py --log2T 19 --samples 192 --table bf16 --rays $r --steps 300 2>> gpurun_out/ab_libs.err | tail -n 1 | sed "s/^{/{\"variant\": \"$v\", /" >> $OUT
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/ab_libs.jsonl"):
    d = json.loads(l)
    print(d["variant"], d["rays_per_step"], d["ms_per_step"], d["kernels_ms_per_step"])
PY
