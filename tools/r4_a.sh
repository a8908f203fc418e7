#!/bin/bash
# round-4 run A: GPU test suite on the 8-byte-record scatter, A/B against the 12-byte records of round 3 (flag 2048), seed study of the
# final PSNR on the reference's schedule.  Every step writes under gpurun_out/r4a/.
OUT=gpurun_out/r4a
mkdir -p $OUT
timeout -k 10 420 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log
tail -3 $OUT/tests.log
for f in 0 2048; do
  timeout -k 10 60 python tools/step_bench.py --rays 1024 --steps 300 --flags $f >> $OUT/ab_scatter.jsonl 2>> $OUT/ab.err || exit 1
  timeout -k 10 60 python tools/step_bench.py --rays 16384 --steps 30 --flags $f >> $OUT/ab_scatter.jsonl 2>> $OUT/ab.err || exit 1
  timeout -k 10 60 python tools/step_bench.py --rays 65536 --steps 10 --flags $f >> $OUT/ab_scatter.jsonl 2>> $OUT/ab.err || exit 1
done
timeout -k 10 60 python tools/step_bench.py --rays 32768 --samples 320 --log2T 22 --table fp16 --steps 10 --flags 0 >> $OUT/ab_scatter.jsonl 2>> $OUT/ab.err
timeout -k 10 60 python tools/step_bench.py --rays 32768 --samples 320 --log2T 22 --table fp16 --steps 10 --flags 2048 >> $OUT/ab_scatter.jsonl 2>> $OUT/ab.err
cat $OUT/ab_scatter.jsonl | cut -c1-420
timeout -k 10 500 python tools/precision_grid.py --epochs 1500 --combos bf16:bf16,fp32:fp32 --seeds 1,2,0 --out $OUT/seed_study.jsonl > $OUT/seed_study.log 2>&1
echo seed study exit $?
