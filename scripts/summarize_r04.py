"""Condense the passes of scripts/profile_r04.sh (gpurun_out/prof_r04/{c2,c3,c5,hbm32,hbm128}/{kt,fetch,write,l2}, probe_*.txt) into
profiles/<tag>_<cfg>_summary.json (+ kernel-stats CSVs) and the tracked per-geometry records profiles/traffic_<name>.json that
bench.py attaches to a run ONLY when that run executed the same kernel geometry:

    {kernel, geometry, workload, head, kernel_sources_sha16, hbm_bytes_per_launch, hbm_read_bytes_per_launch, hbm_write_bytes_per_launch, l2_hit_rate,
     body_us, boundary_us, period_us_unprofiled, rocprof_avg_duration_us, source}

body_us / boundary_us: the -DKMC_PROBE build's in-kernel s_memrealtime stamps (first wave in .. last store issued; gap to the next
launch's first wave).  For the HBM-resident shapes (kernels of ~100 us, which the tool does not distort) body_us is the kernel
trace's mean duration and boundary_us = unprofiled period - that.  Run in the build container (git is here, not on the GPU box).
Usage: python scripts/summarize_r04.py [tag]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import summarize_r02 as base          # trace_stats, counters, bench_line, runcfg_us, one
from bench import kernel_geometry, kernel_sources_sha16, launches_per_generation     # the same matcher / source hash bench.py applies

SRC = os.path.join(ROOT, "gpurun_out", "prof_r04")
DST = os.path.join(ROOT, "profiles")
# cfg dir -> (record name = other_configs key lower-cased, walkers, ndim, workload)
SHAPES = {"c2": ("c2", 65536, 32, "65536 x 32 GaussianIso, moments on"),
          "c3": ("c3", 16384, 64, "16384 x 64 Rosenbrock, moments on"),
          "c5": ("c5", 8192, 1024, "8192 x 1024 GaussianIso, moments on"),
          "hbm32": ("hbm_2mx32", 2097152, 32, "2097152 x 32 GaussianIso, moments on (state 512 MiB)"),
          "hbm128": ("hbm_512kx128", 524288, 128, "524288 x 128 GaussianIso, moments on (state 512 MiB)")}


def head():
    """The commit the passes ran on: scripts/profile_passes.sh refuses a dirty tree and writes it into the snapshot; the box-side script copies it beside its output."""
    p = os.path.join(SRC, "head.txt")
    if not os.path.exists(p):
        raise SystemExit(f"{p} is missing: the passes were not started by scripts/profile_passes.sh (which refuses a tree with uncommitted kernel / bench edits)")
    return open(p).read().strip()


def probe(cfg):
    p = os.path.join(SRC, f"probe_{cfg.upper()}.txt")
    if not os.path.exists(p):
        return None
    recs = [json.loads(l[len("PROBE_JSON "):]) for l in open(p) if l.startswith("PROBE_JSON ")]
    recs = [r for r in recs if r["moments"] == 1]
    return recs[0] if recs else None


def describe_of(path):
    if not path or not os.path.exists(path):
        return None
    for line in open(path):
        if ("half_step_" in line or "generation_" in line) and "grid" in line:
            return line.strip()
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    hd = head()
    for cfg, (name, nw, nd, workload) in SHAPES.items():
        b = os.path.join(SRC, cfg)
        tr = base.one(os.path.join(b, "kt", "**", "*kernel_trace.csv"))
        if not tr:
            print(f"{cfg}: no kernel trace under {b}")
            continue
        cmd = ("python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-island" if cfg == "c2"
               else f"python3 scripts/run_cfg.py {cfg.upper()} <gens> 1")
        out = {"tag": tag, "config": cfg.upper(), "shape": f"{nw} walkers x {nd} dims", "head": hd,
               "command": f"rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}; PMC: the same command under --pmc <group> --kernel-trace, "
                          "one group per pass (scripts/profile_r04.sh)"}
        out.update(base.trace_stats(tr))
        ks = base.one(os.path.join(b, "kt", "**", "*kernel_stats.csv"))
        if ks:
            import csv
            import shutil
            shutil.copy(ks, os.path.join(DST, f"{tag}_{cfg}_kernel_stats.csv"))
            for r in csv.DictReader(open(ks)):
                if r["Name"] == out["kernel_name"]:
                    out["stats_csv"] = {"calls": int(r["Calls"]), "avg_duration_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                        "max_ns": float(r["MaxNs"]), "pct_of_gpu_time": float(r["Percentage"])}
        if cfg == "c2":
            prof, plain = base.bench_line(os.path.join(b, "kt.json")), base.bench_line(os.path.join(SRC, "c2_unprofiled.json"))
            inprof = prof["roofline"]["avg_launch_us"] if prof else None
            live = plain["roofline"]["avg_launch_us"] if plain else None
            how_prof = prof["config"]["execution"] if prof else None
            how = plain["config"]["execution"] if plain else None
        else:
            inprof, live = base.runcfg_us(os.path.join(b, "kt.txt")), base.runcfg_us(os.path.join(b, "unprofiled.txt"))
            how_prof, how = describe_of(os.path.join(b, "kt.txt")), describe_of(os.path.join(b, "unprofiled.txt"))
        out["execution_unprofiled"], out["execution_in_profiled_run"] = how, how_prof
        out["geometry"] = kernel_geometry(how)
        hs = 2 if launches_per_generation(how) == 1 else 1          # half-steps a launch carries (one launch per generation: 2)
        out["half_steps_per_launch"] = hs
        if hs == 2:                                                 # (run_cfg.py prints microseconds per half-step)
            live, inprof = (live * 2 if live else live), (inprof * 2 if inprof else inprof)
        out["hip_event_us_per_launch_unprofiled"], out["hip_event_us_per_launch_in_profiled_run"] = live, inprof
        pm = {}
        for sub in ("fetch", "write", "l2"):
            p = base.one(os.path.join(b, sub, "**", "*counter_collection.csv"))
            if p:
                pm.update(base.counters(p, out["kernel_name"]))
        out["pmc_per_launch"] = pm
        b_read, b_write = (2 * nd + 1) * 8, (nd + 1) * 8
        out["algorithmic_read_bytes_per_launch"] = hs * (nw // 2) * b_read
        out["algorithmic_write_bytes_per_launch_if_all_accepted"] = hs * (nw // 2) * b_write
        if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
            # steady state: the second half of the dispatches (the timed piece); the first half holds warm-up and calibration
            fetch = 2.0 * pm["FETCH_SIZE"]["second_half_mean"] * 1024.0
            write = pm["WRITE_SIZE"]["second_half_mean"] * 1024.0
            out["hbm_read_bytes_per_launch_corrected"], out["hbm_write_bytes_per_launch"] = fetch, write
            out["hbm_bytes_per_launch"] = fetch + write
            out["read_traffic_over_algorithmic_read"] = fetch / out["algorithmic_read_bytes_per_launch"]
            out["correction"] = ("read = 2 x FETCH_SIZE [KiB] x 1024 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md:298); write = WRITE_SIZE [KiB] x 1024; "
                                 "means over the second half of the dispatches")
        if "TCC_HIT_sum" in pm:
            h, m = pm["TCC_HIT_sum"]["second_half_mean"], pm["TCC_MISS_sum"]["second_half_mean"]
            out["l2_hit_rate"] = h / (h + m)
        dur = out["duration_us"]["mean"]
        pr = probe(cfg)
        if pr is not None and pr.get("geometry") == out["geometry"]:
            body, boundary, src = pr["body_us"], pr["boundary_us"], f"-DKMC_PROBE build, gpurun_out/prof_r04/probe_{cfg.upper()}.txt (profiles/{tag}_probe_timeline.txt)"
            out["probe"] = pr
        elif live and live > 30.0 and dur <= 1.15 * live:
            # (the HBM-resident launches are bimodal from process to process -- 98-100 or 105-109 us at 2 097 152 x 32 --: the traced process and the
            #  unprofiled one of a set of passes may sit in different modes, profiles/NOTES.md)
            body, boundary, src = dur, max(live - dur, 0.0), "kernel trace mean duration (a ~100 us kernel: the tool's per-dispatch cost is < 2 %); boundary = unprofiled period - duration"
        else:
            body = boundary = None
            src = "none on record (probe geometry differs or trace distorted)"
        if live:
            alg = out["algorithmic_read_bytes_per_launch"]
            out["fractions"] = {"period_us_unprofiled": live, "algorithmic_read_GBs": alg / live / 1e3, "frac_of_8TBs": alg / live / 1e3 / 8000.0,
                                "frac_of_6.29TBs_measured_copy": alg / live / 1e3 / 6290.0,
                                "body_frac_of_8TBs": (alg / body / 1e3 / 8000.0) if body else None,
                                "pmc_read_GBs_over_period": (out.get("hbm_read_bytes_per_launch_corrected", 0) / live / 1e3) or None,
                                "pmc_total_GBs_over_period": (out.get("hbm_bytes_per_launch", 0) / live / 1e3) or None}
        json.dump(out, open(os.path.join(DST, f"{tag}_{cfg}_summary.json"), "w"), indent=1)
        rec = {"kernel": out["kernel_name"], "geometry": out["geometry"], "workload": workload, "head": hd, "kernel_sources_sha16": kernel_sources_sha16(),
               "hbm_bytes_per_launch": out.get("hbm_bytes_per_launch"), "hbm_read_bytes_per_launch": out.get("hbm_read_bytes_per_launch_corrected"),
               "hbm_write_bytes_per_launch": out.get("hbm_write_bytes_per_launch"), "l2_hit_rate": out.get("l2_hit_rate"),
               "half_steps_per_launch": hs, "body_us": body, "boundary_us": boundary, "body_boundary_source": src, "period_us_unprofiled": live,
               "rocprof_avg_duration_us": dur, "source": f"profiles/{tag}_{cfg}_summary.json"}
        tpath = os.path.join(DST, f"traffic_{name}.json")
        if os.path.exists(tpath):                     # (the periods other boxes measured for this geometry stay on record: scripts/recompute_roofline.py)
            old = json.load(open(tpath))
            if "period_us_unprofiled_other_runs" in old and old.get("geometry") == rec.get("geometry"):
                rec["period_us_unprofiled_other_runs"] = old["period_us_unprofiled_other_runs"]
        json.dump(rec, open(tpath, "w"), indent=1)
        print(json.dumps({k: out.get(k) for k in ("config", "geometry", "hbm_bytes_per_launch", "read_traffic_over_algorithmic_read", "l2_hit_rate", "fractions")}, indent=1))
        print(json.dumps(rec))
    # the HBM-resident shapes in one place (what VERDICT r03 asked for by this name): counters, hit rate, fractions against both peaks
    hbm = {"tag": tag, "head": hd, "what": "the two ensembles of bench.py's other_configs whose state (512 MiB) does not fit the 256 MiB Infinity Cache: "
                                            "same kernels, exact partner rule, moments on; per launch = per half-step",
           "peaks_GBs": {"hbm_spec": 8000.0, "measured_copy": 6290.0}, "shapes": {}}
    for cfg in ("hbm32", "hbm128"):
        path = os.path.join(DST, f"{tag}_{cfg}_summary.json")
        if not os.path.exists(path):
            continue
        o = json.load(open(path))
        fr = o.get("fractions", {})
        pm = o.get("pmc_per_launch", {})
        hbm["shapes"][SHAPES[cfg][0]] = {
            "shape": o["shape"], "kernel": o["kernel_name"], "geometry": o["geometry"],
            "FETCH_SIZE_KiB_per_launch": pm.get("FETCH_SIZE", {}).get("second_half_mean"), "WRITE_SIZE_KiB_per_launch": pm.get("WRITE_SIZE", {}).get("second_half_mean"),
            "TCC_HIT_sum_per_launch": pm.get("TCC_HIT_sum", {}).get("second_half_mean"), "TCC_MISS_sum_per_launch": pm.get("TCC_MISS_sum", {}).get("second_half_mean"),
            "l2_hit_rate": o.get("l2_hit_rate"),
            "hbm_read_bytes_per_launch_corrected": o.get("hbm_read_bytes_per_launch_corrected"), "hbm_write_bytes_per_launch": o.get("hbm_write_bytes_per_launch"),
            "algorithmic_read_bytes_per_launch": o["algorithmic_read_bytes_per_launch"], "read_traffic_over_algorithmic_read": o.get("read_traffic_over_algorithmic_read"),
            "hip_event_us_per_launch_unprofiled": o.get("hip_event_us_per_launch_unprofiled"), "kernel_trace_mean_duration_us": o["duration_us"]["mean"],
            "algorithmic_read_GBs": fr.get("algorithmic_read_GBs"), "frac_of_8.0_TBs": fr.get("frac_of_8TBs"), "frac_of_6.29_TBs": fr.get("frac_of_6.29TBs_measured_copy"),
            "counter_read_plus_write_GBs": fr.get("pmc_total_GBs_over_period"),
            "counter_read_plus_write_over_6.29_TBs": (fr.get("pmc_total_GBs_over_period") or 0) / 6290.0 or None,
            "full_summary": f"profiles/{tag}_{cfg}_summary.json", "record_bench_attaches": f"profiles/traffic_{SHAPES[cfg][0]}.json"}
    if hbm["shapes"]:
        json.dump(hbm, open(os.path.join(DST, f"{tag}_hbm_summary.json"), "w"), indent=1)
    # the probe timelines, as text
    with open(os.path.join(DST, f"{tag}_probe_timeline.txt"), "w") as f:
        for cfg in ("C2", "C3", "C5"):
            p = os.path.join(SRC, f"probe_{cfg}.txt")
            if os.path.exists(p):
                f.write(open(p).read())


if __name__ == "__main__":
    main()
