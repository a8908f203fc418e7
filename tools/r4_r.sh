#!/bin/bash
# how many warm-up steps the 20-step timed region of the driver's command needs to reach the sustained rate (same box, alternating)
mkdir -p gpurun_out/r4r; : > gpurun_out/r4r/warmup.jsonl
C="--cpu-seconds 0 --sub-records 0 --psnr-seconds 0 --full-schedule 0"
for rep in 1 2; do
  for w in 5 50 500 5000; do
    timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup $w $C 2>> gpurun_out/r4r/err | tail -n 1 >> gpurun_out/r4r/warmup.jsonl || exit 1
  done
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 2000 --warmup 100 $C 2>> gpurun_out/r4r/err | tail -n 1 >> gpurun_out/r4r/warmup.jsonl || exit 1
done
python - <<'PY'
import json
for line in open('gpurun_out/r4r/warmup.jsonl'):
    d = json.loads(line)
    print(d['steps'], d['warmup'], round(d['ms_per_step'], 4), round(d['value'] / 1e6, 3), (d.get('sustained') or {}).get('ms_per_step'))
PY
