#!/bin/bash
# the split form of the level-parallel step (each rank's rays in two halves): tests, then one rank's kernels with and without the split
mkdir -p gpurun_out/r4p
timeout -k 10 400 python -m pytest tests/test_hip_levels.py -x -q -m gpu > gpurun_out/r4p/tests.log 2>&1; rc=$?
echo "pytest exit $rc" >> gpurun_out/r4p/tests.log; tail -5 gpurun_out/r4p/tests.log
[ $rc = 0 ] || exit $rc
: > gpurun_out/r4p/emulation.jsonl
for n in 8 4 2; do
  for mode in "" "--split" "" "--split"; do
    timeout -k 10 120 python tools/levels_emulate.py --ranks $n --steps 12 $mode >> gpurun_out/r4p/emulation.jsonl 2>> gpurun_out/r4p/emulation.err || exit 1
  done
done
python - <<'PY'
import json
for line in open('gpurun_out/r4p/emulation.jsonl'):
    d = json.loads(line)
    print(d['ranks'], 'split' if d['rays_in_halves'] else 'whole', d['per_rank_kernels_total_ms'], d['per_rank_kernel_ms'], d['steps'][-1]['table_max_abs_diff'], d['steps'][-1]['acc_max_abs_diff'])
PY
