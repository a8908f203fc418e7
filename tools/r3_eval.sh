#!/bin/bash
# round-3 evaluation-side runs: the single fused gather+MLP kernel against the two-kernel path, with and without
# the feature store, plus rocprofv3 kernel stats and FETCH/WRITE bytes of both; then the time-to-PSNR grid.
export TMPDIR=/tmp
OUT=gpurun_out/r3e
mkdir -p $OUT
rm -f $OUT/eval.jsonl
for f in "" "--store-features" "--two-kernel" "--two-kernel --two-gathers" "--precision fp32"; do
  timeout -k 10 200 python tools/eval_bench.py $f >> $OUT/eval.jsonl 2>> $OUT/eval.err
done
echo eval done
for v in fused two; do
  flag=""; [ $v = two ] && flag="--two-kernel"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$v -o st -- python3 tools/eval_bench.py $flag > $OUT/eval_${v}_under_rocprof.json 2> $OUT/st_$v.err
  cp $(find $OUT/st_$v -name "*kernel_stats.csv" | head -n 1) $OUT/eval_${v}_kernel_stats.csv; rm -rf $OUT/st_$v
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${v}_$c -o pmc -- python3 tools/eval_bench.py $flag > /dev/null 2> $OUT/pmc_${v}_$c.err
    python tools/pmc_summary.py $(find $OUT/pmc_${v}_$c -name "*counter_collection.csv") --json $OUT/eval_${v}_$c.json > $OUT/eval_${v}_$c.txt 2>&1
    rm -rf $OUT/pmc_${v}_$c
  done
done
echo eval profiles done
rm -f $OUT/psnr_race_grid.jsonl
timeout -k 10 900 python tools/psnr_race.py --configs 1024:1e-3,1024:2e-3,1024:4e-3,1024:8e-3,2048:4e-3,4096:4e-3,4096:8e-3,16384:4e-3,16384:8e-3,65536:8e-3 --max-train-s 10 --out $OUT/psnr_race_grid.jsonl > $OUT/race.log 2>&1
echo race done
