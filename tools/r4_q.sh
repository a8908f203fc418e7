#!/bin/bash
# level-parallel emulation at foot_50's shapes (T = 2^22 fp16, S = 320, 8 virtual ranks x 1 024 rays): gradient blocks read in place / gather pass
mkdir -p gpurun_out/r4q; : > gpurun_out/r4q/foot.jsonl
for mode in "" "--gather-pass" "" "--gather-pass"; do
  timeout -k 10 300 python tools/levels_emulate.py --ranks 8 --rays 1024 --log2T 22 --samples 320 --table fp16 --steps 6 $mode >> gpurun_out/r4q/foot.jsonl 2>> gpurun_out/r4q/err || exit 1
done
python - <<'PY'
import json
for line in open('gpurun_out/r4q/foot.jsonl'):
    d = json.loads(line)
    k = d['per_rank_kernel_ms']
    print(d['gradient_blocks'], d['per_rank_kernels_total_ms'], {a: k[a] for a in k if 'scatter' in a or 'gather' in a}, d['steps'][-1]['table_max_abs_diff'], d['steps'][-1]['adam_tail_fused'])
PY
