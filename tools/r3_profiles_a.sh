#!/bin/bash
# round-3 collection A: rocprofv3 kernel stats of the default bench (YAML step) and of the 65 536-ray step, then the cache-counter passes
export TMPDIR=/tmp
OUT=gpurun_out/r3p
mkdir -p $OUT
B="--cpu-seconds 0 --sub-records 0 --psnr-seconds 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1k -o stats -- python3 bench.py --steps 200 --warmup 20 $B > $OUT/bench_1024_under_rocprof.json 2> $OUT/stats1k.err
cp $(find $OUT/stats1k -name "*kernel_stats.csv" | head -n 1) $OUT/kernel_stats_bf16_1024rays.csv; rm -rf $OUT/stats1k
echo stats 1024 done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats64k -o stats -- python3 bench.py --steps 10 --warmup 2 --rays 65536 $B > $OUT/bench_65536_under_rocprof.json 2> $OUT/stats64k.err
cp $(find $OUT/stats64k -name "*kernel_stats.csv" | head -n 1) $OUT/kernel_stats_bf16_65536rays.csv; rm -rf $OUT/stats64k
echo stats 65536 done
bash tools/collect_cache_counters.sh
cp -r gpurun_out/cache $OUT/cache
echo all done
