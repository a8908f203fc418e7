#!/bin/bash
# after the seed fix changed the kernel-source fingerprint: the FETCH_SIZE / WRITE_SIZE passes and kernel stats again (as part A of
# tools/collect_profiles.sh takes them), profiles/pmc_traffic.json from them, then the driver's bench command carrying roofline.traffic
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT gpurun_out/profiles_staged; mkdir -p $OUT gpurun_out/profiles_staged gpurun_out/r4u
COMMON="--cpu-seconds 0 --sub-records 0 --psnr-seconds 0 --full-schedule 0"
for R in 1024 65536; do
  if [ $R = 1024 ]; then ARGS="--steps 200 --warmup 20 --rays $R $COMMON"; PMC="--steps 50 --warmup 10 --rays $R $COMMON"; else ARGS="--steps 10 --warmup 2 --rays $R $COMMON"; PMC="--steps 3 --warmup 1 --rays $R $COMMON"; fi
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$R -o fetch -- python3 bench.py $PMC > $OUT/bench_fetch_$R.json 2> $OUT/fetch_$R.err || exit 1
  echo fetch $R done
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$R -o write -- python3 bench.py $PMC > $OUT/bench_write_$R.json 2> $OUT/write_$R.err || exit 1
  echo write $R done
done
timeout -k 10 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_1024 -o stats -- python3 bench.py --steps 200 --warmup 20 --rays 1024 $COMMON > $OUT/bench_under_rocprof_1024.json 2> $OUT/stats_1024.err || exit 1
echo stats done
NAF_PROFILES_DST=gpurun_out/profiles_staged python tools/install_profiles.py round4 > gpurun_out/r4u/install.log 2>&1
[ -f gpurun_out/profiles_staged/pmc_traffic.json ] || { tail -5 gpurun_out/r4u/install.log; exit 1; }
cp gpurun_out/profiles_staged/pmc_traffic.json profiles/pmc_traffic.json
find $OUT -name '*kernel_stats.csv' -exec cp {} gpurun_out/r4u/stats_1024_kernel_stats.csv \;
rm -rf $OUT/fetch_1024 $OUT/fetch_65536 $OUT/write_1024 $OUT/write_65536 $OUT/stats_1024
timeout -k 10 160 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4u/bench_driver_form.json 2> gpurun_out/r4u/bench.err; rc=$?; echo "bench exit $rc"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4u/bench_driver_form.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
PY
