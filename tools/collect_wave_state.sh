#!/bin/bash
# Where the waves of every kernel of the step spend their cycles (MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_WAIT_ANY = parked at
# s_waitcnt / a barrier, SQ_WAIT_INST_ANY = issue stalls, SQ_ACTIVE_INST_* = issuing; the three classes add up to SQ_WAVE_CYCLES.
# One counter pass per batch size; summaries land in gpurun_out/wave/.
export TMPDIR=/tmp
OUT=gpurun_out/wave
mkdir -p $OUT
for R in 1024 65536; do
  if [ $R = 1024 ]; then A="--steps 50 --warmup 10"; else A="--steps 3 --warmup 1"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
    --output-format csv -d $OUT/p_$R -o p -- python3 bench.py $A --rays $R --cpu-seconds 0 --sub-records 0 --psnr-seconds 0 --full-schedule 0 > $OUT/bench_$R.json 2> $OUT/p_$R.err \
    && python tools/pmc_summary.py $(find $OUT/p_$R -name "*counter_collection.csv") --json $OUT/wave_state_$R.json > $OUT/wave_state_$R.txt 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT \
    --output-format csv -d $OUT/q_$R -o q -- python3 bench.py $A --rays $R --cpu-seconds 0 --sub-records 0 --psnr-seconds 0 --full-schedule 0 > /dev/null 2> $OUT/q_$R.err \
    && python tools/pmc_summary.py $(find $OUT/q_$R -name "*counter_collection.csv") --json $OUT/wave_insts_$R.json > $OUT/wave_insts_$R.txt 2>&1
  rm -rf $OUT/p_$R $OUT/q_$R
done
cat $OUT/wave_state_1024.txt $OUT/wave_state_65536.txt
