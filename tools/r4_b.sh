#!/bin/bash
# round-4 run B: where the two scatter kernels spend their time -- per-level HIP-event timings and SQ counters for the 8-byte-record
# kernels (flags 0) and the round-3 kernels (flags 2048), then a same-box A/B of library variants.
export TMPDIR=/tmp
OUT=gpurun_out/r4b
mkdir -p $OUT
B="--cpu-seconds 0 --sub-records 0 --psnr-seconds 0 --full-schedule 0"
for F in 0 2048; do
  timeout -k 10 120 python bench.py --rays 65536 --steps 5 --warmup 2 $B --cfg-flags $F --per-level > $OUT/per_level_65536_f$F.json 2>> $OUT/err.log || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
    --output-format csv -d $OUT/p_$F -o p -- python3 bench.py --steps 3 --warmup 1 --rays 65536 $B --cfg-flags $F > /dev/null 2>> $OUT/err.log \
    && python tools/pmc_summary.py $(find $OUT/p_$F -name "*counter_collection.csv") --json $OUT/wave_state_65536_f$F.json > $OUT/wave_state_65536_f$F.txt 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR \
    --output-format csv -d $OUT/q_$F -o q -- python3 bench.py --steps 3 --warmup 1 --rays 65536 $B --cfg-flags $F > /dev/null 2>> $OUT/err.log \
    && python tools/pmc_summary.py $(find $OUT/q_$F -name "*counter_collection.csv") --json $OUT/wave_insts_65536_f$F.json > $OUT/wave_insts_65536_f$F.txt 2>&1
  rm -rf $OUT/p_$F $OUT/q_$F
done
cat $OUT/wave_state_65536_f*.txt $OUT/wave_insts_65536_f*.txt | grep scatter
rm -f gpurun_out/ab_libs.jsonl
RAYS="1024 65536" bash tools/ab_libs.sh "$@"
