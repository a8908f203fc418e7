#!/bin/bash
# round-4 closing run: the GPU suite on the final library, then the profile collection (parts A and C of tools/collect_profiles.sh)
mkdir -p gpurun_out/r4z
rm -rf gpurun_out/prof gpurun_out/profiles_staged
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r4z/tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r4z/tests.log
tail -3 gpurun_out/r4z/tests.log
NAF_TAG=round4 bash tools/collect_profiles.sh A > gpurun_out/r4z/collect_a.log 2>&1; echo "A exit $?"
