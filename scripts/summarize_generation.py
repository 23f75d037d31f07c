"""Condense the passes of scripts/profile_generation.sh (gpurun_out/prof_gen/<cfg>_<one|two>/) into profiles/<tag>_generation_summary.json (python scripts/summarize_generation.py <tag> [cfg ...]):
per job and mode the dominant kernel's dispatch count, duration and begin-to-begin period from the kernel trace (second half of the
dispatches), next to the HIP-event figure of the same profiled process and of an unprofiled one."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import summarize_r04 as base

SRC = os.path.join(ROOT, "gpurun_out", "prof_gen")


def us_of(path):
    if not os.path.exists(path):
        return None, None
    t = open(path).read()
    m = re.search(r"us/half-step ([0-9.]+)", t)
    how = [l.strip() for l in t.splitlines() if "grid" in l]
    return (float(m.group(1)) if m else None), (how[-1] if how else None)


TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
SHAPES = {"MID4K": "4096 walkers x 4 dims", "MID16K": "16384 walkers x 4 dims", "MID8Kx64": "8192 walkers x 64 dims", "MID16Kx32": "16384 walkers x 32 dims",
          "C3": "16384 walkers x 64 dims, chained Rosenbrock"}
CFGS = sys.argv[2:] or ["MID4K", "MID16K"]
out = {"tag": TAG, "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 scripts/run_cfg.py <cfg> 4096 1 (KMC_DEBUG=fused=1 / fused=0 for the one- / two-launch rows); "
                              "PMC: the same under --pmc <group> --kernel-trace, one group per pass", "jobs": {}}
for cfg in CFGS:
    shape = SHAPES[cfg]
    for mode in ("one", "two"):
        tr = glob.glob(os.path.join(SRC, f"{cfg}_{mode}", "**", "*kernel_trace.csv"), recursive=True)
        if not tr:
            continue
        base.KEY = "generation_" if mode == "one" else "half_step"
        st = base.trace_stats(tr[0])
        st.pop("excerpt_12_consecutive_dispatches", None)
        us_prof, how = us_of(os.path.join(SRC, f"{cfg}_{mode}_kt.txt"))
        us_plain, _ = us_of(os.path.join(SRC, f"{cfg}_{mode}_unprofiled.txt"))
        per_gen = 1 if mode == "one" else 2
        pmc = {}
        for name in ("fetch", "write", "l2"):
            cs = glob.glob(os.path.join(SRC, f"{cfg}_{mode}_{name}", "**", "*counter_collection.csv"), recursive=True)
            if cs:
                for k, v in base.counters(cs[0], st["kernel_name"]).items():
                    pmc[k] = v["second_half_mean"]
        if pmc:
            # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KiB-like units of 1 KB per count x ... as in scripts/summarize_r04.py: bytes = 1024 x counter, read x 2 on gfx950
            rd = 2.0 * 1024.0 * pmc.get("FETCH_SIZE", 0.0)
            wr = 1024.0 * pmc.get("WRITE_SIZE", 0.0)
            hit, miss = pmc.get("TCC_HIT_sum", 0.0), pmc.get("TCC_MISS_sum", 0.0)
            st["pmc_per_launch"] = {"counters": pmc, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "l2_hit_rate": hit / (hit + miss) if hit + miss > 0 else None}
        st.update({"shape": shape, "launches_per_generation": per_gen, "execution": how,
                   "us_per_half_step_hip_events_profiled_process": us_prof, "us_per_half_step_hip_events_unprofiled": us_plain,
                   "us_per_half_step_from_trace_period": st["period_us_from_trace"]["mean"] * per_gen / 2.0})
        out["jobs"][f"{cfg}_{mode}_launch{'es' if per_gen == 2 else ''}_per_generation"] = st
json.dump(out, open(os.path.join(ROOT, "profiles", f"{TAG}_generation_summary.json"), "w"), indent=1)
for k, v in out["jobs"].items():
    if "pmc_per_launch" in v:
        print("   PMC per launch: read", round(v["pmc_per_launch"]["hbm_read_bytes"] / 1e6, 3), "MB, written", round(v["pmc_per_launch"]["hbm_write_bytes"] / 1e6, 3), "MB, L2 hit", v["pmc_per_launch"]["l2_hit_rate"])
    print(k, v["kernel_name"][:60], "dispatches", v["dispatches"], "duration mean us", round(v["duration_us"]["mean"], 3), "period mean us", round(v["period_us_from_trace"]["mean"], 3),
          "| HIP events us/half-step: profiled", v["us_per_half_step_hip_events_profiled_process"], "unprofiled", v["us_per_half_step_hip_events_unprofiled"])
