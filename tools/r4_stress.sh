#!/bin/bash
# the randomised parity sweeps on the round's final kernels; summaries in gpurun_out/r4s/
mkdir -p gpurun_out/r4s
for t in stress_fused stress_fine_depths stress_entry_points stress_more stress_unfused stress_levels; do
  timeout -k 10 400 python tools/$t.py > gpurun_out/r4s/$t.log 2>&1; echo "$t exit $? : $(tail -n 1 gpurun_out/r4s/$t.log | cut -c1-300)"
done
