"""Condense gpurun_out/prof_isl (rocprofv3 of scripts/run_island.py) into profiles/<tag>_island_*."""
import collections
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", "prof_isl")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01c"
d = collections.defaultdict(list)
for sub in ("sq", "fetch", "write"):
    for r in csv.DictReader(open(os.path.join(base, sub, "c2_counter_collection.csv"))):
        if "island_epoch" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: statistics.mean(v) for k, v in d.items()}
ks = [r for r in csv.DictReader(open(os.path.join(base, "kt", "c2_kernel_stats.csv"))) if "island_epoch" in r["Name"]][0]
summary = {
    "tag": f"{tag}_island", "kernel": ks["Name"], "calls": int(ks["Calls"]), "avg_duration_ns": float(ks["AverageNs"]),
    "generations_per_launch": 64, "walker_steps_per_launch": 65536 * 64,
    "walker_steps_per_s": 65536 * 64 / (float(ks["AverageNs"]) * 1e-9),
    "pmc_per_launch": out,
    "lds_bank_conflict_fraction": out.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, out.get("SQ_LDS_IDX_ACTIVE", 1)),
    "valu_active_fraction_of_wave_cycles": out.get("SQ_ACTIVE_INST_VALU", 0) / max(1, out.get("SQ_WAVE_CYCLES", 1)),
    "hbm_bytes_per_launch": 2 * out.get("FETCH_SIZE", 0) * 1024 + out.get("WRITE_SIZE", 0) * 1024,
    "command": "rocprofv3 --kernel-trace --stats / --pmc <counters> --kernel-trace -- python3 scripts/run_island.py 256 64 {2048|1024}",
}
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_island_summary.json"), "w"), indent=1)
shutil.copy(os.path.join(base, "kt", "c2_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_island_kernel_stats.csv"))
print(json.dumps(summary, indent=1))
