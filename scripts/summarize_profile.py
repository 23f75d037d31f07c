"""Condense rocprofv3 output (gpurun_out/prof/{kt,fetch,write,l2}) into profiles/<tag>_*.{csv,json}.
Usage: python scripts/summarize_profile.py r01a [kernel-substring]"""
import collections
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def counters(path, key):
    d = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(path)):
        if key in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d, dur


def main():
    global SRC
    tag = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "half_step_vec"
    if len(sys.argv) > 3:                     # alternative rocprofv3 output directory under gpurun_out/
        SRC = os.path.join(ROOT, "gpurun_out", sys.argv[3])
    write_traffic = len(sys.argv) <= 3
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "kt", "c2_kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats.csv"))
    out = {"tag": tag, "kernel": key,
           "command": "rocprofv3 --kernel-trace --stats / --pmc <counter> --kernel-trace -- python3 bench.py --steps {2|1} --no-cpu-baseline"}
    for r in csv.DictReader(open(os.path.join(SRC, "kt", "c2_kernel_stats.csv"))):
        if key in r["Name"]:
            out["kernel_name"] = r["Name"]
            out["calls"] = int(r["Calls"])
            out["avg_duration_ns"] = float(r["AverageNs"])
            out["min_duration_ns"] = float(r["MinNs"])
            out["pct_of_gpu_time"] = float(r["Percentage"])
    pm = {}
    for sub in ("fetch", "write", "l2"):
        p = os.path.join(SRC, sub, "c2_counter_collection.csv")
        if not os.path.exists(p):
            continue
        d, dur = counters(p, key)
        for name, v in d.items():
            n = len(v)
            pm[name] = {"n": n, "mean": statistics.mean(v), "burnin_half_mean": statistics.mean(v[: n // 2]),
                        "sampling_half_mean": statistics.mean(v[n // 2:])}
    out["pmc_per_launch"] = pm
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
        # reads (MI355X_MICROARCH.md, HBM section) -> double it.  WRITE_SIZE is exact for 16-B stores.
        fetch = 2.0 * pm["FETCH_SIZE"]["mean"] * 1024.0
        write = pm["WRITE_SIZE"]["mean"] * 1024.0
        out["hbm_bytes_per_launch"] = fetch + write
        out["hbm_read_bytes_per_launch_corrected"] = fetch
        out["hbm_write_bytes_per_launch"] = write
        out["correction"] = "read = 2 x FETCH_SIZE KiB x 1024 (gfx950 halves wide coalesced reads); write = WRITE_SIZE KiB x 1024"
    if "TCC_HIT_sum" in pm:
        h, m = pm["TCC_HIT_sum"]["mean"], pm["TCC_MISS_sum"]["mean"]
        out["l2_hit_rate"] = h / (h + m)
    json.dump(out, open(os.path.join(DST, f"{tag}_summary.json"), "w"), indent=1)
    if write_traffic:
        json.dump({"hbm_bytes_per_launch": out.get("hbm_bytes_per_launch"), "source": f"profiles/{tag}_summary.json"},
                  open(os.path.join(DST, "traffic_c2.json"), "w"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
