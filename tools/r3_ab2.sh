#!/bin/bash
# window gathers vs two gathers per batch size and table size
OUT=${OUT:-gpurun_out/r3n}
mkdir -p $OUT; rm -f $OUT/ab2.jsonl
for a in "--log2T 19 --samples 192 --table bf16 --rays 2048" "--log2T 19 --samples 192 --table bf16 --rays 4096" "--log2T 19 --samples 192 --table bf16 --rays 16384" "--log2T 19 --samples 192 --table bf16 --rays 65536" "--log2T 20 --samples 192 --table bf16 --rays 65536" "--log2T 21 --samples 192 --table bf16 --rays 32768" "--log2T 22 --samples 320 --table fp16 --rays 32768" "--log2T 22 --samples 320 --table fp32 --rays 16384" "--log2T 19 --samples 192 --table fp32 --rays 16384" "--log2T 19 --samples 576 --table bf16 --rays 16384"; do
  for f in 0 32; do
    timeout -k 10 100 python tools/step_bench.py $a --flags $f 2>> $OUT/ab2.err | tail -n 1 >> $OUT/ab2.jsonl
  done
done
echo ab2 done
