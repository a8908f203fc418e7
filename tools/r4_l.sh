#!/bin/bash
# four waves per ray in the MLP forward at small batches: tests, same-box A/B
mkdir -p gpurun_out/r4l
timeout -k 10 600 python -m pytest tests/test_hip_fused.py tests/test_hip_training.py tests/test_hip_edge_cases.py tests/test_hip_forward_paths.py tests/test_hip_levels.py tests/test_hip_configs.py -x -q -m gpu > gpurun_out/r4l/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4l/tests.log
[ $rc = 0 ] || exit $rc
rm -f gpurun_out/ab_libs.jsonl
RAYS="128 512 1024 2048" bash tools/ab_libs.sh base pro base pro > gpurun_out/r4l/ab.txt 2>&1
cat gpurun_out/r4l/ab.txt
