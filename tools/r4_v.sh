#!/bin/bash
# the GPU test files tools/r4_s.sh and tools/r4_t.sh did not cover, on the seed-fix build (together: the whole GPU suite)
mkdir -p gpurun_out/r4v
timeout -k 10 215 python -m pytest tests/test_hip_hash.py tests/test_hip_render_ops.py tests/test_hip_configs.py tests/test_hip_forward_paths.py -x -q -m gpu > gpurun_out/r4v/tests.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/r4v/tests.log
tail -3 gpurun_out/r4v/tests.log
exit $rc
