#!/bin/bash
# round-4 run C: GPU test suite on the current library, then same-box A/B of library variants, then SQ counters of the current one
export TMPDIR=/tmp
OUT=gpurun_out/r4c
mkdir -p $OUT
L=neuralvolumetricreconstructionformedicalimages_amd/lib
cp $L/libnaf_hip.so $OUT/../.current.so
timeout -k 10 420 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log
tail -4 $OUT/tests.log
rm -f gpurun_out/ab_libs.jsonl
RAYS="1024 16384 65536" bash tools/ab_libs.sh "$@" | tee $OUT/ab.txt
cp $OUT/../.current.so $L/libnaf_hip.so
B="--cpu-seconds 0 --sub-records 0 --psnr-seconds 0 --full-schedule 0"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
  --output-format csv -d $OUT/p -o p -- python3 bench.py --steps 3 --warmup 1 --rays 65536 $B > /dev/null 2>> $OUT/err.log \
  && python tools/pmc_summary.py $(find $OUT/p -name "*counter_collection.csv") --json $OUT/wave_state_65536.json > $OUT/wave_state_65536.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR \
  --output-format csv -d $OUT/q -o q -- python3 bench.py --steps 3 --warmup 1 --rays 65536 $B > /dev/null 2>> $OUT/err.log \
  && python tools/pmc_summary.py $(find $OUT/q -name "*counter_collection.csv") --json $OUT/wave_insts_65536.json > $OUT/wave_insts_65536.txt 2>&1
rm -rf $OUT/p $OUT/q
cat $OUT/wave_state_65536.txt $OUT/wave_insts_65536.txt | grep scatter
