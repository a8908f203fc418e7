#!/bin/bash
# MLP prologue / fold restructuring: tests, same-box A/B, phase stamps
mkdir -p gpurun_out/r4k
L=neuralvolumetricreconstructionformedicalimages_amd/lib
timeout -k 10 500 python -m pytest tests/test_hip_fused.py tests/test_hip_training.py tests/test_hip_edge_cases.py tests/test_hip_forward_paths.py tests/test_hip_levels.py -x -q -m gpu > gpurun_out/r4k/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4k/tests.log
[ $rc = 0 ] || exit $rc
rm -f gpurun_out/ab_libs.jsonl
RAYS="128 1024 65536" bash tools/ab_libs.sh base pro base pro > gpurun_out/r4k/ab.txt 2>&1
cat gpurun_out/r4k/ab.txt
cp $L/ab/stamps.so $L/libnaf_hip.so
for r in 128 1024 4096; do timeout -k 10 100 python tools/mlp_stamps.py --rays $r 2>/dev/null | tail -n 1 >> gpurun_out/r4k/stamps.jsonl; done; cat gpurun_out/r4k/stamps.jsonl
