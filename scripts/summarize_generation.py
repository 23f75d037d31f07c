"""Condense the passes of scripts/profile_generation.sh (gpurun_out/prof_gen/<cfg>_<one|two>/) into profiles/r04_generation_summary.json:
per job and mode the dominant kernel's dispatch count, duration and begin-to-begin period from the kernel trace (second half of the
dispatches), next to the HIP-event figure of the same profiled process and of an unprofiled one."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import summarize_r02 as base

SRC = os.path.join(ROOT, "gpurun_out", "prof_gen")


def us_of(path):
    if not os.path.exists(path):
        return None, None
    t = open(path).read()
    m = re.search(r"us/half-step ([0-9.]+)", t)
    how = [l.strip() for l in t.splitlines() if "grid" in l]
    return (float(m.group(1)) if m else None), (how[-1] if how else None)


out = {"tag": "r04", "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 scripts/run_cfg.py <MID4K|MID16K> 4096 1 (KMC_DEBUG=fused=0 for the two-launch rows)", "jobs": {}}
for cfg, shape in (("MID4K", "4096 walkers x 4 dims"), ("MID16K", "16384 walkers x 4 dims")):
    for mode in ("one", "two"):
        tr = glob.glob(os.path.join(SRC, f"{cfg}_{mode}", "**", "*kernel_trace.csv"), recursive=True)
        if not tr:
            continue
        base.KEY = "generation_lane" if mode == "one" else "half_step"
        st = base.trace_stats(tr[0])
        st.pop("excerpt_12_consecutive_dispatches", None)
        us_prof, how = us_of(os.path.join(SRC, f"{cfg}_{mode}_kt.txt"))
        us_plain, _ = us_of(os.path.join(SRC, f"{cfg}_{mode}_unprofiled.txt"))
        per_gen = 1 if mode == "one" else 2
        st.update({"shape": shape, "launches_per_generation": per_gen, "execution": how,
                   "us_per_half_step_hip_events_profiled_process": us_prof, "us_per_half_step_hip_events_unprofiled": us_plain,
                   "us_per_half_step_from_trace_period": st["period_us_from_trace"]["mean"] * per_gen / 2.0})
        out["jobs"][f"{cfg}_{mode}_launch{'es' if per_gen == 2 else ''}_per_generation"] = st
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_generation_summary.json"), "w"), indent=1)
for k, v in out["jobs"].items():
    print(k, v["kernel_name"][:60], "dispatches", v["dispatches"], "duration mean us", round(v["duration_us"]["mean"], 3), "period mean us", round(v["period_us_from_trace"]["mean"], 3),
          "| HIP events us/half-step: profiled", v["us_per_half_step_hip_events_profiled_process"], "unprofiled", v["us_per_half_step_hip_events_unprofiled"])
