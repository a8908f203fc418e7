#!/bin/bash
# (historical: the A/B of commit f5437ab, naf_render_train_adam_pipelined -- the entry point was removed afterwards; profiles/round4_ab_pipeline_across_steps.jsonl)
mkdir -p gpurun_out/r4h
timeout -k 10 500 python -m pytest tests/test_hip_render_ops.py tests/test_abi_symbols.py -x -q -m gpu > gpurun_out/r4h/tests.log 2>&1; rc=$?; tail -15 gpurun_out/r4h/tests.log
[ $rc = 0 ] || exit $rc
C="--cpu-seconds 0 --sub-records 0 --psnr-seconds 0 --full-schedule 0 --steps 3000 --warmup 200"
rm -f gpurun_out/r4h/ab.jsonl
for v in none 8-16 none 8-16 4-12 0-8 12-16 none; do
  if [ $v = none ]; then A=""; else A="--pipeline-levels $v"; fi
  timeout -k 10 200 python bench.py $C $A 2>> gpurun_out/r4h/ab.err | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'pipeline':'$v','ms_per_step':d['ms_per_step'],'sustained':d.get('sustained'),'kernels':d.get('kernels_ms_per_step'),'final_loss':d.get('final_loss')}))" >> gpurun_out/r4h/ab.jsonl
done
cat gpurun_out/r4h/ab.jsonl
