#!/bin/bash
# the from-blocks scatter's publish-once seed fix: the level-parallel and distributed GPU tests, then smoke()
mkdir -p gpurun_out/r4s
timeout -k 10 240 python -m pytest tests/test_hip_levels.py tests/test_hip_dist.py -x -q -m gpu > gpurun_out/r4s/tests.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/r4s/tests.log
tail -3 gpurun_out/r4s/tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4s/smoke.log 2>&1; rc=$?; echo "smoke exit $rc"; tail -n 2 gpurun_out/r4s/smoke.log
exit $rc
