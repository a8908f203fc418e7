#!/bin/bash
# in-place gradient blocks (scatter_v2.h, GradBlocks): the level-parallel tests, then the one-process emulation with and without the
# separate gather pass on the same box
mkdir -p gpurun_out/r4m
timeout -k 10 400 python -m pytest tests/test_hip_levels.py tests/test_hip_dist.py -x -q -m gpu > gpurun_out/r4m/tests.log 2>&1; rc=$?
echo "pytest exit $rc" >> gpurun_out/r4m/tests.log; tail -5 gpurun_out/r4m/tests.log
[ $rc = 0 ] || exit $rc
: > gpurun_out/r4m/emulation.jsonl
for n in 8 4 2; do
  for mode in "" "--gather-pass" "" "--gather-pass"; do
    timeout -k 10 120 python tools/levels_emulate.py --ranks $n --steps 12 $mode >> gpurun_out/r4m/emulation.jsonl 2>> gpurun_out/r4m/emulation.err || exit 1
  done
done
python - <<'PY'
import json
for line in open('gpurun_out/r4m/emulation.jsonl'):
    d = json.loads(line)
    k = d['per_rank_kernel_ms']
    print(d['ranks'], d['gradient_blocks'], d['per_rank_kernels_total_ms'], {a: k[a] for a in k if 'scatter' in a or 'gather' in a}, d['per_rank_phase_ms'].get('scatter_adam'), d['steps'][-1]['table_max_abs_diff'])
PY
