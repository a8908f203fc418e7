#!/bin/bash
# MLP epilogue diet: the bf16 / training tests on the new library, then a same-box A/B against the previous build
mkdir -p gpurun_out/r4g
rm -f gpurun_out/ab_libs.jsonl
timeout -k 10 500 python -m pytest tests/test_hip_fused.py tests/test_hip_training.py tests/test_hip_edge_cases.py tests/test_hip_forward_paths.py -x -q -m gpu > gpurun_out/r4g/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4g/tests.log
[ $rc = 0 ] || exit $rc
RAYS="1024 16384 65536" bash tools/ab_libs.sh base fwd4 base fwd4 > gpurun_out/r4g/ab.txt 2>&1
cat gpurun_out/r4g/ab.txt
