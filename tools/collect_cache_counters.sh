#!/bin/bash
# L1 / L2 request counters of the chest step's kernels and the FETCH_SIZE calibration for scattered gathers
# (item 4c).  Run through gpurun from the repo root; summaries land in gpurun_out/cache/.
export TMPDIR=/tmp
OUT=gpurun_out/cache
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || echo "rocprofv3 -L failed"
ARGS="--steps 3 --warmup 1 --rays 65536 --cpu-seconds 0 --sub-records 0 --psnr-seconds 0"
pass() {   # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 bench.py $ARGS > $OUT/bench_$name.json 2> $OUT/$name.err \
    && python tools/pmc_summary.py $(find $OUT/$name -name "*counter_collection.csv") --json $OUT/$name.json > $OUT/$name.txt 2>&1 \
    || echo "pass $name failed (see $OUT/$name.err)"
  rm -rf $OUT/$name
}
pass tcp_a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tcp_b TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp_c TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pass tcc_a TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass tcc_b TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass tcc_c TCC_READ_sum TCC_TAG_STALL_sum
echo "step passes done"
# FETCH_SIZE against a known number of distinct 64-byte sectors
./tools/bin/gather_fetch_calib 2048 > $OUT/calib_plain.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o calib -- ./tools/bin/gather_fetch_calib 2048 > $OUT/calib_fetch.txt 2> $OUT/calib_fetch.err \
  && python - <<'PY' > $OUT/calib_fetch_summary.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/cache/calib_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Kernel_Name"][:40], r["Counter_Name"], float(r["Counter_Value"]), "raw ->", float(r["Counter_Value"]) * 1024 / 2**20, "MiB")
PY
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/calib_tcc -o calib -- ./tools/bin/gather_fetch_calib 2048 > $OUT/calib_tcc.txt 2> $OUT/calib_tcc.err \
  && python - <<'PY' > $OUT/calib_tcc_summary.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/cache/calib_tcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Kernel_Name"][:40], r["Counter_Name"], float(r["Counter_Value"]))
PY
rm -rf $OUT/calib_fetch $OUT/calib_tcc
echo "calibration done"
