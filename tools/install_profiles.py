#!/usr/bin/env python3
"""Copy the summaries produced by tools/collect_profiles.sh (gpurun_out/prof/) into profiles/ under their round-1 names
and derive profiles/pmc_traffic.json (HBM bytes per step and kernel group, read by bench.py for `roofline.traffic`)."""
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "prof")
DST = os.environ.get("NAF_PROFILES_DST", os.path.join(REPO, "profiles"))      # the GPU box stages into gpurun_out/ (no 64 MiB of traces)
os.makedirs(DST, exist_ok=True)
TAG = sys.argv[1] if len(sys.argv) > 1 else "round4"


def copy(src, dst):
    if not os.path.exists(os.path.join(SRC, src)):       # the other part of tools/collect_profiles.sh produces it
        return
    shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, dst))
    print("profiles/" + dst)


import glob  # noqa: E402

for R in (1024, 65536):
    hits = glob.glob(os.path.join(SRC, f"stats_{R}", "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copyfile(hits[0], os.path.join(DST, f"{TAG}_kernel_stats_bf16_{R}rays.csv"))
        print(f"profiles/{TAG}_kernel_stats_bf16_{R}rays.csv")
    copy(f"bench_under_rocprof_{R}.json", f"{TAG}_bench_under_rocprof_{R}rays.json")
copy("bench_default.json", f"{TAG}_bench_default.json")
copy("batch_sweep.jsonl", f"{TAG}_batch_sweep.jsonl")
copy("bench_fp32_16384.json", f"{TAG}_bench_fp32_16384rays.json")
copy("bench_per_level_16384.json", f"{TAG}_bench_per_level_16384rays.json")
for src, dst in (("bench_force_dp.json", "bench_data_parallel_step_one_gpu.json"), ("bench_interleaved.json", "bench_levels_interleaved_encoder.json"), ("eval.jsonl", "eval_throughput.jsonl"),
                 ("train_py.json", "train_py_throughput.json"), ("shapes.jsonl", "other_shapes_step_times.jsonl"),
                 ("levels_emulation.jsonl", "level_parallel_emulation.jsonl"), ("bench_level_parallel_one_rank.json", "bench_level_parallel_one_rank.json"),
                 ("psnr_16384_bf16.json", "chest_psnr_vs_time_16384rays.json"), ("psnr_16384_fp32.json", "chest_psnr_vs_time_16384rays_fp32.json"),
                 ("psnr_1024_bf16.json", "chest_psnr_vs_time_1024rays.json"), ("psnr_race_grid.jsonl", "psnr_race_grid.jsonl")):
    if os.path.exists(os.path.join(SRC, src)):
        copy(src, f"{TAG}_{dst}")

sys.path.insert(0, REPO)
from neuralvolumetricreconstructionformedicalimages_amd.build import source_fingerprint  # noqa: E402

# HBM-side bytes per launch of every kernel of the step, per batch size.  MI355X_MICROARCH.md: value * 1024 (done by pmc_summary),
# FETCH_SIZE doubled on gfx950 (it tallies 64 B per fabric read request, and a streamed line is one 128-byte request;
# profiles/round3_cache_counters.md shows the calibration: exact for scattered single-sector gathers, half for whole lines --
# every kernel of this step streams whole lines except the encoder, whose fetches are 1 % of its traffic at T = 2^19).
# Keys are the names bench.py's profiler uses (naf_profile_collect): the 16-point MFMA kernels report under the generic names.
ALIAS = {"mlp16_forward_kernel": "mlp_forward_kernel", "mlp16_backward_kernel": "mlp_backward_kernel",
         "scatter_bin2_kernel": "scatter_bin_kernel", "scatter_reduce2_kernel": "scatter_reduce_kernel"}      # scatter_v2.h
by_rays = {}
for R in (1024, 65536):
    fetch = glob.glob(os.path.join(SRC, f"fetch_{R}", "**", "*counter_collection.csv"), recursive=True)
    write = glob.glob(os.path.join(SRC, f"write_{R}", "**", "*counter_collection.csv"), recursive=True)
    if not (fetch and write):
        continue
    per_dispatch = os.path.join(DST, f"{TAG}_pmc_bytes_per_dispatch_{R}rays.json")
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), fetch[0], write[0], "--json", per_dispatch], check=True)
    pd = json.load(open(per_dispatch))
    table = {}
    for k, row in pd.items():
        name = ALIAS.get(k, k)
        table[name] = table.get(name, 0.0) + 2.0 * row.get("FETCH_SIZE", 0.0) + row.get("WRITE_SIZE", 0.0)
    by_rays[str(R)] = {"bf16": table}
traffic = {
    "csrc_fingerprint": source_fingerprint(),      # bench.py prints these figures only while the kernel sources are unchanged
    "collected_at_commit": os.environ.get("NAF_COMMIT") or subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
    "_comment": "HBM-side bytes per LAUNCH and kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of "
                "`python3 bench.py --rays R`, value*1024, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; per-dispatch "
                f"figures in profiles/{TAG}_pmc_bytes_per_dispatch_<R>rays.json, calibration in profiles/round3_cache_counters.md",
    "by_rays": by_rays,
}
if by_rays:                                            # (part B of tools/collect_profiles.sh has no counter passes)
    json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
    print("profiles/pmc_traffic.json", {r: {k: round(v / 1e6, 1) for k, v in t["bf16"].items()} for r, t in by_rays.items()})


# ---- T = 2^22: HBM bytes fetched by the fused forward's encoder ----------------------------------------------------------
t22 = {"_comment": "foot_50 shapes (L=16, T=2^22, S=320, 32768 rays = 10.49 M points per launch): encode_kernel of the fused training step. "
                   "FETCH_SIZE from `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/step_bench.py --log2T 22 --samples 320 --table <t> --rays 32768 --steps 3` "
                   "(value * 1024; random 4/8-byte gathers fetch 64-byte sectors, so no x2 correction is applied: the doubled figure is listed "
                   "separately), kernel time of the same launches WITHOUT counters from the plain run"}
for t in ("fp16", "fp32"):
    csv_path = os.path.join(SRC, f"t22_{t}", "t22_counter_collection.csv")
    plain = os.path.join(SRC, f"t22_{t}_plain.json")
    if not (os.path.exists(csv_path) and os.path.exists(plain)):
        continue
    import csv as _csv
    vals = [float(r["Counter_Value"]) * 1024 for r in _csv.DictReader(open(csv_path))
            if "encode_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    run = json.loads(open(plain).read().strip().splitlines()[-1])
    ms = run["kernels_ms_per_step"]["encode_kernel"]
    fetched = sum(vals) / max(len(vals), 1)
    t22[t] = {"table_MB": run["table_MB"], "encode_kernel_ms": ms, "dispatches_counted": len(vals), "fetch_bytes_per_launch": fetched,
              "hbm_read_TBps": fetched / (ms * 1e-3) / 1e12, "fraction_of_8TBps": fetched / (ms * 1e-3) / 8e12,
              "fetch_bytes_per_launch_if_doubled": 2 * fetched, "step_ms": run["ms_per_step"], "step_kernels_ms": run["kernels_ms_per_step"]}
if len(t22) > 1:
    json.dump(t22, open(os.path.join(DST, f"{TAG}_hash_forward_T22_fetch.json"), "w"), indent=1)
    print(f"profiles/{TAG}_hash_forward_T22_fetch.json", {k: (round(v["hbm_read_TBps"], 2), v["encode_kernel_ms"]) for k, v in t22.items() if k != "_comment"})

# ---- SQ instruction mix and MFMA utilisation -------------------------------------------------------------------------
import collections  # noqa: E402
import csv  # noqa: E402
import re  # noqa: E402


def counters(*files):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        path = os.path.join(SRC, f)
        if not os.path.exists(path):
            continue
        for r in csv.DictReader(open(path)):
            name = re.sub(r"[<(].*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("naf::", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


sq = counters("sqa/sqa_counter_collection.csv", "sqb/sqb_counter_collection.csv")
mf = counters("mfma/mfma_counter_collection.csv")
kernels = ("encode_kernel", "mlp16_forward_kernel", "mlp16_backward_kernel", "mlp_forward_kernel", "mlp_backward_kernel", "scatter_bin_kernel", "scatter_reduce_kernel", "scatter_bin2_kernel", "scatter_reduce2_kernel", "draw_scan_rays_kernel", "adam_kernel")
out = ["# Counter evidence per kernel (MI355X, chest_50 bf16, 65 536 rays/step = 12.58 M points; tools/collect_profiles.sh)", "",
       "## Dynamic instruction mix", "",
       "`rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -- python3 bench.py --steps 2 --warmup 1 --rays 65536 --cpu-seconds 0`,",
       "second pass `SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY`.", "",
       "| kernel | waves | VALU / wave | SALU / wave | LDS / wave | VMEM rd / wave | VMEM wr / wave | VALU issue time (4 cyc x insts / 1024 SIMDs @ 2.4 GHz) | LDS bank-conflict cycles / CU |",
       "|---|---|---|---|---|---|---|---|---|"]
for k in kernels:
    row = sq.get(k)
    if not row or "SQ_WAVES" not in row:
        continue
    w = row["SQ_WAVES"]
    out.append(f"| {k} | {w:.0f} | {row['SQ_INSTS_VALU'] / w:.0f} | {row['SQ_INSTS_SALU'] / w:.0f} | {row['SQ_INSTS_LDS'] / w:.0f} | "
               f"{row['SQ_INSTS_VMEM_RD'] / w:.1f} | {row['SQ_INSTS_VMEM_WR'] / w:.1f} | {row['SQ_INSTS_VALU'] * 4 / 1024 / 2.4e6:.2f} ms | "
               f"{row.get('SQ_LDS_BANK_CONFLICT', 0) / 256 / 1e6:.2f} M |")
out += ["", "## MFMA utilisation of the MLP kernels", "",
        "`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -- python3 bench.py --steps 3 --warmup 1 --rays 65536 --cpu-seconds 0`", "",
        "| kernel | SQ_INSTS_MFMA | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CU_CYCLES | MFMA busy / (4 SIMD x CU busy) |", "|---|---|---|---|---|"]
for k in ("mlp16_forward_kernel", "mlp16_backward_kernel", "mlp_forward_kernel", "mlp_backward_kernel"):
    row = mf.get(k)
    if not row:
        continue
    out.append(f"| {k} | {row['SQ_INSTS_MFMA']:.0f} | {row['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} | {row['SQ_BUSY_CU_CYCLES']:.0f} | "
               f"{100 * row['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * row['SQ_BUSY_CU_CYCLES']):.1f} % |")
if sq or mf:
    open(os.path.join(DST, f"{TAG}_sq_counters.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))
