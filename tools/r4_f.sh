#!/bin/bash
OUT=gpurun_out/r4f
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log
tail -4 $OUT/tests.log
timeout -k 10 100 python tools/step_bench.py --rays 32768 --samples 320 --log2T 22 --table fp16 --steps 5 --flags 1 > $OUT/foot_per_level.json 2>> $OUT/err.log
timeout -k 10 200 python tools/levels_emulate.py > $OUT/levels_emulate.jsonl 2>> $OUT/err.log; tail -3 $OUT/levels_emulate.jsonl | cut -c1-600
timeout -k 10 120 python tools/standalone_bench.py > $OUT/standalone.json 2>> $OUT/err.log; cat $OUT/standalone.json
