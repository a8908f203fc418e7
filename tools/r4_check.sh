#!/bin/bash
# sanity run of a restored tree: the GPU suite, then the default bench line
mkdir -p gpurun_out/r4chk
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r4chk/tests.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/r4chk/tests.log
tail -3 gpurun_out/r4chk/tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r4chk/bench_default.json 2> gpurun_out/r4chk/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4chk/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('kernels_ms_per_step'), d['roofline']['frac'])
print({k: v.get('value') for k, v in d.get('sub_records', {}).items()} if isinstance(d.get('sub_records'), dict) else d.get('sub_records'))
PY
