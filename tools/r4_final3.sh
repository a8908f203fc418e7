#!/bin/bash
# round-4 closing run after the in-place gradient blocks: profile collection part A on the final library (the GPU suite ran on it in the
# call before: tools/r4_check.sh), then the level-parallel emulation lines of part C
mkdir -p gpurun_out/r4z
rm -rf gpurun_out/prof gpurun_out/profiles_staged
NAF_TAG=round4 bash tools/collect_profiles.sh A > gpurun_out/r4z/collect_a.log 2>&1; echo "A exit $?"
OUT=gpurun_out/prof
rm -f $OUT/levels_emulation.jsonl
for n in 2 4 8; do
  timeout -k 10 200 python tools/levels_emulate.py --ranks $n 2>> $OUT/levels.err | tail -n 1 >> $OUT/levels_emulation.jsonl
done
timeout -k 10 200 python bench.py --force-dp --dp-mode levels --steps 1000 --psnr-seconds 0 --cpu-seconds 0 --sub-records 0 --full-schedule 0 2> $OUT/levels_one_rank.err | tail -n 1 > $OUT/bench_level_parallel_one_rank.json
NAF_PROFILES_DST=gpurun_out/profiles_staged python tools/install_profiles.py round4 > gpurun_out/r4z/install_c.log 2>&1
ls gpurun_out/profiles_staged
