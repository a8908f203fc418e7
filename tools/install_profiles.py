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
DST = os.path.join(REPO, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "round1"


def copy(src, dst):
    shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, dst))
    print("profiles/" + dst)


copy("stats/stats_kernel_stats.csv", f"{TAG}_kernel_stats_bf16_65536rays.csv")
copy("bench_under_rocprof.json", f"{TAG}_bench_under_rocprof.json")
copy("bench_default.json", f"{TAG}_bench_default.json")
copy("batch_sweep.jsonl", f"{TAG}_batch_sweep.jsonl")
copy("bench_fp32_16384.json", f"{TAG}_bench_fp32_16384rays.json")
copy("bench_per_level_16384.json", f"{TAG}_bench_per_level_16384rays.json")

per_dispatch = os.path.join(DST, f"{TAG}_pmc_bytes_per_dispatch.json")
subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"),
                os.path.join(SRC, "fetch", "fetch_counter_collection.csv"),
                os.path.join(SRC, "write", "write_counter_collection.csv"), "--json", per_dispatch], check=True)
pd = json.load(open(per_dispatch))


def hbm(kernel):   # MI355X_MICROARCH.md: value * 1024 (done by pmc_summary), FETCH_SIZE doubled on gfx950 for wide reads
    row = pd.get(kernel, {})
    return 2.0 * row.get("FETCH_SIZE", 0.0) + row.get("WRITE_SIZE", 0.0)


traffic = {
    "_comment": "HBM-side bytes per training step (65536 rays, 12.58 M points) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                "(separate runs), value*1024, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; per-dispatch "
                f"figures in profiles/{TAG}_pmc_bytes_per_dispatch.json, calibration notes in DESIGN.md section 5",
    "bf16": {
        "hash_forward": hbm("encode_kernel"),
        "hash_backward": hbm("scatter_bin_kernel") + hbm("scatter_reduce_kernel"),
        "mlp_forward": hbm("mlp_forward_kernel"),
        "mlp_backward": hbm("mlp_backward_kernel"),
    },
}
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
print("profiles/pmc_traffic.json", {k: round(v / 1e9, 2) for k, v in traffic["bf16"].items()})
