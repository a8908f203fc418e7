#!/bin/bash
# Collects the evidence kept under profiles/ on the GPU box (run through gpurun from the repo root):
#   rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE counter passes (separate runs), the default bench line, a batch
#   sweep, the fp32 mode, the per-level split, the data-parallel step on one GPU, evaluation and train.py throughput,
#   other BASELINE shapes and the chest PSNR curves.  Everything lands in gpurun_out/prof/; tools/install_profiles.py copies
#   the summaries into profiles/.
# Two parts, one gpurun call each (a call is capped at 20 minutes):  bash tools/collect_profiles.sh A   /   ... B
set -e
export TMPDIR=/tmp
PART=${1:-A}
OUT=gpurun_out/prof
mkdir -p $OUT
if [ "$PART" = A ]; then
COMMON="--cpu-seconds 0 --sub-records 0 --psnr-seconds 0 --full-schedule 0"
# the bench's default workload (chest_50.yaml's own step: 1 024 rays) and the throughput end of the batch curve (65 536 rays)
for R in 1024 65536; do
  if [ $R = 1024 ]; then ARGS="--steps 200 --warmup 20 --rays $R $COMMON"; PMC="--steps 50 --warmup 10 --rays $R $COMMON"; else ARGS="--steps 10 --warmup 2 --rays $R $COMMON"; PMC="--steps 3 --warmup 1 --rays $R $COMMON"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$R -o stats -- python3 bench.py $ARGS > $OUT/bench_under_rocprof_$R.json 2> $OUT/stats_$R.err
  echo stats $R done
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$R -o fetch -- python3 bench.py $PMC > $OUT/bench_fetch_$R.json 2> $OUT/fetch_$R.err
  echo fetch $R done
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$R -o write -- python3 bench.py $PMC > $OUT/bench_write_$R.json 2> $OUT/write_$R.err
done
ARGS="--steps 10 --warmup 2 --rays 65536 $COMMON"
PMC="--steps 3 --warmup 1 --rays 65536 $COMMON"
echo write done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/mfma -o mfma -- python3 bench.py $PMC > $OUT/bench_mfma.json 2> $OUT/mfma.err
echo mfma done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sqa -o sqa -- python3 bench.py --steps 2 --warmup 1 --rays 65536 $COMMON > $OUT/bench_sqa.json 2> $OUT/sqa.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $OUT/sqb -o sqb -- python3 bench.py --steps 2 --warmup 1 --rays 65536 $COMMON > $OUT/bench_sqb.json 2> $OUT/sqb.err
echo sq done
# the counter passes are in: derive profiles/pmc_traffic.json now, so that the default line below carries `roofline.traffic`
NAF_PROFILES_DST=gpurun_out/profiles_staged python tools/install_profiles.py ${NAF_TAG:-round4} > $OUT/install_a.log 2>&1
cp gpurun_out/profiles_staged/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo default done
for r in 128 256 512 1024 2048 4096 16384 65536 262144 1048576; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --rays $r $COMMON >> $OUT/batch_sweep.jsonl 2>> $OUT/batch_sweep.err
done
echo sweep done
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --rays 16384 --precision fp32 $COMMON > $OUT/bench_fp32_16384.json 2> $OUT/fp32.err
timeout -k 10 300 python bench.py --per-level --steps 5 --warmup 2 --rays 16384 $COMMON > $OUT/bench_per_level_16384.json 2> $OUT/per_level.err
timeout -k 10 300 python bench.py --force-dp --steps 200 --warmup 20 --cpu-seconds 0 --psnr-seconds 0 --full-schedule 0 > $OUT/bench_force_dp.json 2> $OUT/force_dp.err
# the encoder with all levels of a point tile in flight at once (what a single fused gather+MLP kernel would do to the caches)
timeout -k 10 300 python bench.py --interleaved-levels --rays 65536 --steps 10 --warmup 3 $COMMON > $OUT/bench_interleaved.json 2> $OUT/interleaved.err
echo modes done
NAF_PROFILES_DST=gpurun_out/profiles_staged python tools/install_profiles.py ${NAF_TAG:-round4}
rm -rf $OUT/stats_1024 $OUT/stats_65536 $OUT/fetch_1024 $OUT/fetch_65536 $OUT/write_1024 $OUT/write_65536 $OUT/mfma $OUT/sqa $OUT/sqb
ls gpurun_out/profiles_staged
exit 0
fi
if [ "$PART" = C ]; then
# ---- part C (round 4): the short half of part B -- evaluation, train.py's loop, the other BASELINE shapes, level-parallel emulation
rm -f $OUT/eval.jsonl $OUT/shapes.jsonl $OUT/levels_emulation.jsonl
for f in "" "--precision fp32"; do
  timeout -k 10 200 python tools/eval_bench.py $f >> $OUT/eval.jsonl 2>> $OUT/eval.err
done
timeout -k 10 300 python tools/train_throughput.py 2> $OUT/train_py.err | grep "^{" | tail -n 1 > $OUT/train_py.json
for a in "--log2T 22 --samples 320 --table fp16 --rays 32768" "--log2T 21 --samples 192 --table bf16 --rays 32768" "--log2T 20 --samples 192 --table bf16 --rays 65536" "--log2T 19 --samples 576 --table bf16 --rays 16384" "--log2T 19 --samples 192 --table bf16 --rays 65536" "--log2T 19 --samples 320 --table bf16 --rays 1024" "--log2T 19 --samples 576 --table bf16 --rays 1024 --steps 100"; do
  timeout -k 10 100 python tools/step_bench.py $a 2>> $OUT/shapes.err | tail -n 1 >> $OUT/shapes.jsonl
done
for n in 2 4 8; do
  timeout -k 10 200 python tools/levels_emulate.py --ranks $n 2>> $OUT/levels.err | tail -n 1 >> $OUT/levels_emulation.jsonl
done
timeout -k 10 300 python tools/levels_emulate.py --ranks 8 --rays 1024 --log2T 22 --samples 320 --table fp16 2>> $OUT/levels.err | tail -n 1 >> $OUT/levels_emulation.jsonl
timeout -k 10 200 python bench.py --force-dp --dp-mode levels --steps 1000 --psnr-seconds 0 --cpu-seconds 0 --sub-records 0 --full-schedule 0 2> $OUT/levels_one_rank.err | tail -n 1 > $OUT/bench_level_parallel_one_rank.json
NAF_PROFILES_DST=gpurun_out/profiles_staged python tools/install_profiles.py ${NAF_TAG:-round4}
ls gpurun_out/profiles_staged
exit 0
fi
# ---- part B: evaluation side, train.py loop, other BASELINE shapes, T = 2^22 fetch bytes, PSNR curves and the time-to-PSNR grid ----
rm -f $OUT/eval.jsonl $OUT/shapes.jsonl $OUT/psnr_race_grid.jsonl
for f in "" "--precision fp32" "--fused" "--fused --store-features"; do
  timeout -k 10 200 python tools/eval_bench.py $f >> $OUT/eval.jsonl 2>> $OUT/eval.err
done
timeout -k 10 300 python tools/train_throughput.py 2> $OUT/train_py.err | grep "^{" | tail -n 1 > $OUT/train_py.json
for a in "--log2T 22 --samples 320 --table fp16 --rays 32768" "--log2T 21 --samples 192 --table bf16 --rays 32768" "--log2T 20 --samples 192 --table bf16 --rays 65536" "--log2T 19 --samples 576 --table bf16 --rays 16384" "--log2T 19 --samples 192 --table bf16 --rays 65536"; do
  timeout -k 10 100 python tools/step_bench.py $a 2>> $OUT/shapes.err | tail -n 1 >> $OUT/shapes.jsonl
done
echo eval + shapes done
# level-parallel step at chest size, N virtual ranks in one process: parity with the single-GPU step and one rank's share of the kernels
rm -f $OUT/levels_emulation.jsonl
for n in 2 4 8; do
  timeout -k 10 200 python tools/levels_emulate.py --ranks $n 2>> $OUT/levels.err | tail -n 1 >> $OUT/levels_emulation.jsonl
done
timeout -k 10 200 python tools/levels_emulate.py --ranks 8 --default-buckets 2>> $OUT/levels.err | tail -n 1 >> $OUT/levels_emulation.jsonl
timeout -k 10 200 python tools/levels_emulate.py --ranks 8 --precision fp32 2>> $OUT/levels.err | tail -n 1 >> $OUT/levels_emulation.jsonl
for r in 1024 4096; do      # foot_50 shapes
  timeout -k 10 300 python tools/levels_emulate.py --ranks 8 --rays $r --log2T 22 --samples 320 --table fp16 2>> $OUT/levels.err | tail -n 1 >> $OUT/levels_emulation.jsonl
done
timeout -k 10 200 python bench.py --force-dp --dp-mode levels --steps 1000 --psnr-seconds 0 --cpu-seconds 0 --sub-records 0 --full-schedule 0 2> $OUT/levels_one_rank.err | tail -n 1 > $OUT/bench_level_parallel_one_rank.json
echo level-parallel emulation done
# T = 2^22 (foot_50 shapes, the table is larger than every cache level below the Infinity Cache): HBM bytes fetched by the
# encoder of the FUSED forward, quoted against the north star's 60 % bar in DESIGN.md section 4.1
for t in fp16 fp32; do
  timeout -k 10 200 python tools/step_bench.py --log2T 22 --samples 320 --table $t --rays 32768 2>> $OUT/shapes.err | tail -n 1 > $OUT/t22_${t}_plain.json
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/t22_$t -o t22 -- python3 tools/step_bench.py --log2T 22 --samples 320 --table $t --rays 32768 --steps 3 > $OUT/t22_${t}_pmc.json 2> $OUT/t22_$t.err
done
echo T22 fetch done
timeout -k 10 300 python tools/train_chest.py --rays 16384 --steps 2500 --eval-every 500 --out $OUT/psnr_16384_bf16.json > $OUT/psnr_a.log 2>&1
timeout -k 10 300 python tools/train_chest.py --rays 16384 --steps 2500 --eval-every 500 --precision fp32 --out $OUT/psnr_16384_fp32.json > $OUT/psnr_b.log 2>&1
timeout -k 10 300 python tools/train_chest.py --rays 1024 --steps 20000 --eval-every 5000 --out $OUT/psnr_1024_bf16.json > $OUT/psnr_c.log 2>&1
# time to 30 / 35 / 38 dB volume PSNR per operating point (rays per step : learning rate); 1024:1e-3 is the reference's own
timeout -k 10 400 python tools/psnr_race.py --configs 256:2e-3,512:2e-3,512:4e-3,1024:1e-3,1024:2e-3,1024:4e-3,1024:8e-3,2048:4e-3,4096:4e-3,16384:4e-3,65536:8e-3 --max-train-s 10 --out $OUT/psnr_race_grid.jsonl > $OUT/race.log 2>&1
echo all done
# summarise on the box and keep only the summaries (the kernel traces alone exceed what gpurun copies back)
NAF_PROFILES_DST=gpurun_out/profiles_staged python tools/install_profiles.py ${NAF_TAG:-round4}
rm -rf $OUT/t22_fp16 $OUT/t22_fp32
ls gpurun_out/profiles_staged
