#!/bin/bash
OUT=gpurun_out/r4e
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log
tail -4 $OUT/tests.log
for r in 1024 16384 65536; do timeout -k 10 60 python tools/step_bench.py --rays $r --steps 100 >> $OUT/step.jsonl; done
timeout -k 10 60 python tools/step_bench.py --rays 32768 --samples 320 --log2T 22 --table fp16 --steps 10 >> $OUT/step.jsonl
cut -c1-330 $OUT/step.jsonl
timeout -k 10 120 python tools/standalone_bench.py > $OUT/standalone.json 2> $OUT/standalone.err; cat $OUT/standalone.json; tail -3 $OUT/standalone.err
timeout -k 10 120 python tools/standalone_bench.py --dtype fp32 >> $OUT/standalone.json 2>> $OUT/standalone.err; tail -1 $OUT/standalone.json
