#!/bin/bash
# last call of the round: what the driver runs at round end, on the committed tree -- the GPU suite, smoke(), its bench command
mkdir -p gpurun_out/r4last
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r4last/tests.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/r4last/tests.log
tail -3 gpurun_out/r4last/tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4last/smoke.log 2>&1; echo "smoke exit $?"; tail -n 2 gpurun_out/r4last/smoke.log
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4last/bench_driver_form.json 2> gpurun_out/r4last/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4last/bench_driver_form.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['roofline'], d['cpu_baseline']['value'], d['full_schedule']['bf16']['psnr_db'], d['full_schedule']['fp32']['psnr_db'])
PY
