#!/bin/bash
# round-4 closing run, second call: part C of tools/collect_profiles.sh (evaluation, train.py loop, other shapes, level-parallel emulation)
# and the wave-state counter passes
mkdir -p gpurun_out/r4z
NAF_TAG=round4 bash tools/collect_profiles.sh C > gpurun_out/r4z/collect_c.log 2>&1; echo "C exit $?"
bash tools/collect_wave_state.sh > gpurun_out/r4z/wave.log 2>&1; echo "wave exit $?"
python tools/wave_state_md.py gpurun_out/wave > gpurun_out/wave/round4_wave_state.md 2>> gpurun_out/r4z/wave.log; echo "md exit $?"
