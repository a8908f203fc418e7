#!/bin/bash
# after the seed fix, on the committed tree: the driver's bench command, then the single-GPU scatter / fused / training tests
mkdir -p gpurun_out/r4t
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4t/bench_driver_form.json 2> gpurun_out/r4t/bench.err; rc=$?; echo "bench exit $rc"
[ $rc = 0 ] || exit $rc
tail -c 600 gpurun_out/r4t/bench_driver_form.json
timeout -k 10 230 python -m pytest tests/test_hip_fused.py tests/test_hip_training.py tests/test_hip_edge_cases.py -x -q -m gpu > gpurun_out/r4t/tests.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/r4t/tests.log
tail -3 gpurun_out/r4t/tests.log
exit $rc
