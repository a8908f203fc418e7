#!/bin/bash
# round-3 A/B runs of the training step (diagnostic flags); output: $OUT/ab.jsonl (one line per variant)
OUT=${OUT:-gpurun_out/r3c}
mkdir -p $OUT
B="--cpu-seconds 0 --sub-records 0 --psnr-seconds 0"
run() {  # rays steps warm flags...
  rays=$1; steps=$2; warm=$3; shift 3
  python bench.py --rays $rays --steps $steps --warmup $warm $B "$@" 2>> $OUT/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'rays':$rays,'flags':'$*','ms_per_step':round(d['ms_per_step'],4),'sustained_ms':d['sustained'] and d['sustained']['ms_per_step'],'kernels':d['kernels_ms_per_step']}))" >> $OUT/ab.jsonl
}
run 1024 300 30 --precision fp32
run 16384 30 3 --precision fp32
run 65536 10 3 --precision fp32
echo ab done
