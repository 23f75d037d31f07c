"""Condense the passes of scripts/profile_r04.sh (gpurun_out/prof_r04/{c2,c3,c5,hbm32,hbm128}/{kt,fetch,write,l2}, probe_*.txt, probe_light_*.txt, phase_*.txt) into
profiles/<tag>_<cfg>_summary.json (+ kernel-stats CSVs, + profiles/<tag>_c2_dispatches.csv: start / end of 2 000 consecutive steady-state dispatches of the headline
kernel) and the tracked per-geometry records profiles/traffic_<name>.json that bench.py attaches to a run ONLY when that run executed the same kernel geometry:

    {kernel, geometry, launch_mode, workload, head, kernel_sources_sha16, hbm_bytes_per_launch, hbm_read_bytes_per_launch, hbm_write_bytes_per_launch, l2_hit_rate,
     body_us, boundary_us, period_us_in_kernel, burnin{...}, period_us_unprofiled, period_us_unprofiled_burnin, rocprof_avg_duration_us, rocprof_inflated,
     duration_us, duration_source, source}

launch_mode: how the launches of every pass were issued (scripts/profile_r04.sh forces the mode the bench line runs the config in; scripts/recompute_roofline.py
refuses a record whose mode differs from the line's).  body_us / boundary_us / period_us_in_kernel: the LIGHT probe build's in-kernel s_memrealtime stamps
(-DKMC_PROBE=2: first wave in .. last store issued; gap to the next launch's first wave; first wave to first wave) of launches that credit moments, `burnin` the
same for launches that do not.  rocprof_avg_duration_us: the kernel trace's mean per-dispatch duration; rocprof_inflated: it exceeds the unprofiled launch period --
a kernel cannot take longer than the period that contains it, so for the 3-6 us launches the tool's figure is NOT evidence (under the tool every dispatch carries a
completion signal and starts when its predecessor retires: `gap_end_to_next_start_us.share_zero`; the traced "duration" is the launch period under the tool).
duration_us / duration_source: the kernel-duration figure of the record -- the trace's where it is not inflated, else the light probe's stamps (first wave in .. last store
issued); the probe build's in-kernel launch period must stay within 5 % of the production build's period of the same passes (scripts/recompute_roofline.py).  Run in the build container (git is here, not on the GPU box).
Usage: python scripts/summarize_r04.py [tag]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_geometry, kernel_sources_sha16, launch_mode_of, launches_per_generation     # the same matchers / source hash bench.py applies

KEY = "half_step_vec"


def one(pattern):
    g = glob.glob(pattern, recursive=True)
    return g[0] if g else None


def pct(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p / 100.0 * len(v)))]


def trace_stats(path):
    rows = [r for r in csv.DictReader(open(path)) if KEY in r["Kernel_Name"] or (KEY == "half_step_vec" and "generation_group" in r["Kernel_Name"])]   # (C3 runs one launch per generation since round 5)
    by = collections.Counter(r["Kernel_Name"] for r in rows)
    name = by.most_common(1)[0][0]
    rows = [r for r in rows if r["Kernel_Name"] == name]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    st = [int(r["Start_Timestamp"]) for r in rows]
    en = [int(r["End_Timestamp"]) for r in rows]
    dur = [e - s for s, e in zip(st, en)]
    per = [b - a for a, b in zip(st, st[1:])]
    gap = [st[i + 1] - en[i] for i in range(len(st) - 1)]
    # steady state: the second half of the dispatches (after warm-up, launch-mode measurement and graph instantiation)
    h = len(per) // 2
    sper, sdur, sgap = per[h:], dur[h:], gap[h:]
    mid = h + len(sper) // 2
    excerpt = [{"i": i, "start_ns": st[i] - st[mid], "end_ns": en[i] - st[mid], "duration_ns": dur[i],
                "gap_to_next_ns": gap[i] if i < len(gap) else None} for i in range(mid, min(mid + 12, len(st)))]
    return {"kernel_name": name, "dispatches": len(rows),
            "duration_us": {"mean": statistics.mean(sdur) / 1e3, "median": statistics.median(sdur) / 1e3, "min": min(sdur) / 1e3,
                            "p10": pct(sdur, 10) / 1e3, "p90": pct(sdur, 90) / 1e3},
            "period_us_from_trace": {"mean": statistics.mean(sper) / 1e3, "median": statistics.median(sper) / 1e3,
                                     "p10": pct(sper, 10) / 1e3, "p90": pct(sper, 90) / 1e3,
                                     "over": f"dispatches {h}..{len(per)} of {len(rows)} (second half: steady state)"},
            "gap_end_to_next_start_us": {"median": statistics.median(sgap) / 1e3, "mean": statistics.mean(sgap) / 1e3,
                                         "share_zero": sum(1 for g in sgap if g <= 0) / len(sgap)},
            "excerpt_12_consecutive_dispatches": excerpt}


def counters(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"] == name:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"n": len(v), "mean": statistics.mean(v), "first_half_mean": statistics.mean(v[: len(v) // 2]),
                "second_half_mean": statistics.mean(v[len(v) // 2:])} for k, v in d.items()}


def bench_line(path):
    if not path or not os.path.exists(path):
        return None
    for line in open(path):
        line = line.strip()
        if line.startswith('{"metric"'):             # (the result line; since round 5 the full record precedes it as {"bench_detail": ...})
            return json.loads(line)
    return None


def runcfg_us(path):
    if not path or not os.path.exists(path):
        return None
    m = re.search(r"us/half-step ([0-9.]+)", open(path).read())
    return float(m.group(1)) if m else None




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
    """{1: record of launches that credit moments, 0: burn-in launches} of the LIGHT probe build (falls back to the full probe's)."""
    for name in (f"probe_light_{cfg.upper()}.txt", f"probe_{cfg.upper()}.txt"):
        p = os.path.join(SRC, name)
        if os.path.exists(p):
            recs = [json.loads(l[len("PROBE_JSON "):]) for l in open(p) if l.startswith("PROBE_JSON ")]
            if recs:
                return {r["moments"]: r for r in recs}
    return {}


def phase_us(cfg, mom, hs):
    """The production build's launch period in one phase (scripts/run_cfg.py <cfg> 2048 <mom>, same launch mode), per launch."""
    v = runcfg_us(os.path.join(SRC, f"phase_{cfg.upper()}_m{mom}.txt"))
    return v * hs if v else None


def describe_of(path):
    if not path or not os.path.exists(path):
        return None
    for line in open(path):
        if ("half_step_" in line or "generation_" in line) and "grid" in line:
            return line.strip()
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    hd = head()
    for cfg, (name, nw, nd, workload) in SHAPES.items():
        b = os.path.join(SRC, cfg)
        tr = one(os.path.join(b, "kt", "**", "*kernel_trace.csv"))
        if not tr:
            print(f"{cfg}: no kernel trace under {b}")
            continue
        cmd = ("python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-island" if cfg == "c2"
               else f"python3 scripts/run_cfg.py {cfg.upper()} <gens> 1")
        out = {"tag": tag, "config": cfg.upper(), "shape": f"{nw} walkers x {nd} dims", "head": hd,
               "command": f"rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}; PMC: the same command under --pmc <group> --kernel-trace, "
                          "one group per pass (scripts/profile_r04.sh)"}
        out.update(trace_stats(tr))
        ks = one(os.path.join(b, "kt", "**", "*kernel_stats.csv"))
        if ks:
            import csv
            import shutil
            shutil.copy(ks, os.path.join(DST, f"{tag}_{cfg}_kernel_stats.csv"))
            for r in csv.DictReader(open(ks)):
                if r["Name"] == out["kernel_name"]:
                    out["stats_csv"] = {"calls": int(r["Calls"]), "avg_duration_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                        "max_ns": float(r["MaxNs"]), "pct_of_gpu_time": float(r["Percentage"])}
        if cfg == "c2":
            prof, plain = bench_line(os.path.join(b, "kt.json")), bench_line(os.path.join(SRC, "c2_unprofiled.json"))
            inprof = prof["roofline"]["avg_launch_us"] if prof else None
            live = plain["roofline"]["avg_launch_us"] if plain else None
            how_prof = prof["config"]["execution"] if prof else None
            how = plain["config"]["execution"] if plain else None
        else:
            inprof, live = runcfg_us(os.path.join(b, "kt.txt")), runcfg_us(os.path.join(b, "unprofiled.txt"))
            how_prof, how = describe_of(os.path.join(b, "kt.txt")), describe_of(os.path.join(b, "unprofiled.txt"))
        out["execution_unprofiled"], out["execution_in_profiled_run"] = how, how_prof
        out["geometry"] = kernel_geometry(how)
        out["launch_mode"], out["launch_mode_in_profiled_run"] = launch_mode_of(how), launch_mode_of(how_prof)
        if cfg == "c2":                      # per-dispatch start / end of the headline kernel, steady state ("keep per-dispatch start/end")
            rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr)) if r["Kernel_Name"] == out["kernel_name"])
            mid = rows[len(rows) * 3 // 4: len(rows) * 3 // 4 + 2000]
            with open(os.path.join(DST, f"{tag}_c2_dispatches.csv"), "w") as fcsv:
                fcsv.write("start_ns,end_ns\n" + "".join(f"{a - mid[0][0]},{b - mid[0][0]}\n" for a, b in mid))
        hs = 2 if launches_per_generation(how) == 1 else 1          # half-steps a launch carries (one launch per generation: 2)
        out["half_steps_per_launch"] = hs
        if hs == 2:                                                 # (run_cfg.py prints microseconds per half-step)
            live, inprof = (live * 2 if live else live), (inprof * 2 if inprof else inprof)
        out["hip_event_us_per_launch_unprofiled"], out["hip_event_us_per_launch_in_profiled_run"] = live, inprof
        pm = {}
        for sub in ("fetch", "write", "l2"):
            p = one(os.path.join(b, sub, "**", "*counter_collection.csv"))
            if p:
                pm.update(counters(p, out["kernel_name"]))
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
        cred, burn = pr.get(1), pr.get(0)
        inflated = bool(live and dur > 1.03 * live and live < 30.0)      # (a ~100 us launch is not inflated by the tool's ~0.8 us: the HBM-resident launches are two-valued by PROCESS, below)
        out["rocprof_inflated"] = inflated
        if inflated:
            out["rocprof_inflated_why"] = (f"the traced mean duration ({dur:.2f} us) exceeds the unprofiled launch period ({live:.2f} us): under rocprofv3 every dispatch carries a completion "
                                           f"signal and its begin stamp is taken when its predecessor retires (gap 0 in {100 * out['gap_end_to_next_start_us']['share_zero']:.0f} % of the dispatches), "
                                           "so the traced duration is the launch PERIOD under the tool (+0.7-0.9 us per dispatch), not the kernel's duration: not evidence for a 3-6 us launch")
        body = boundary = in_kernel = None
        src = "none on record (probe geometry differs or trace distorted)"
        if cred is not None and cred.get("geometry") == out["geometry"]:
            body, boundary, in_kernel = cred["body_us"], cred["boundary_us"], cred["period_us_in_kernel"]
            src = f"-DKMC_PROBE={'2 (light)' if cred.get('probe') == 'light' else '1'} build, launches that credit moments (profiles/{tag}_probe_timeline.txt)"
            out["probe"] = {"credited": cred, "burnin": burn}
            out["probe_launch_mode"] = launch_mode_of(cred.get("execution"))
        elif live and live > 30.0 and dur <= 1.15 * live:
            # (the HBM-resident launches are bimodal from process to process -- 98-100 or 105-109 us at 2 097 152 x 32 --: the traced process and the
            #  unprofiled one of a set of passes may sit in different modes, profiles/NOTES.md)
            body, boundary, src = dur, max(live - dur, 0.0), "kernel trace mean duration (a ~100 us kernel: the tool's per-dispatch cost is < 2 %); boundary = unprofiled period - duration"
        # duration_source, a short enum (the bench line carries it): rocprof_trace | light_probe_stamps | none
        if not inflated:
            duration, dsrc, dnote = dur, "rocprof_trace", "rocprofv3 kernel trace, mean per-dispatch duration"
        elif body is not None:
            duration, dsrc, dnote = body, "light_probe_stamps", ("the -DKMC_PROBE=2 stamps: first wave in .. last store issued -- what a kernel's duration is as seen from inside it (the dispatch's own start / end "
                                                                 "bracket it by the wave-launch ramp and the end-of-kernel write-back, which the stamps count to the boundary); the rocprofv3 duration is "
                                                                 "inflated by the tool (rocprof_inflated_why)")
        else:
            duration, dsrc, dnote = None, "none", "rocprofv3 duration inflated, no probe of this geometry"
        out["duration_source_note"] = dnote
        out["duration_us"]["record"], out["duration_source"] = duration, dsrc
        if live:
            alg = out["algorithmic_read_bytes_per_launch"]
            out["fractions"] = {"period_us_unprofiled": live, "algorithmic_read_GBs": alg / live / 1e3, "frac_of_8TBs": alg / live / 1e3 / 8000.0,
                                "frac_of_6.29TBs_measured_copy": alg / live / 1e3 / 6290.0,
                                "duration_frac_of_8TBs": (alg / duration / 1e3 / 8000.0) if duration else None,
                                "body_frac_of_8TBs": (alg / body / 1e3 / 8000.0) if body else None,
                                "pmc_read_GBs_over_period": (out.get("hbm_read_bytes_per_launch_corrected", 0) / live / 1e3) or None,
                                "pmc_total_GBs_over_period": (out.get("hbm_bytes_per_launch", 0) / live / 1e3) or None}
        json.dump(out, open(os.path.join(DST, f"{tag}_{cfg}_summary.json"), "w"), indent=1)
        per_cred, per_burn = phase_us(cfg, 1, hs), phase_us(cfg, 0, hs)
        if live and live >= 30.0 and dur > 1.03 * live:
            out["two_valued_note"] = (f"the traced process ran these launches at {dur:.1f} us, the unprofiled one at {live:.1f}: the HBM-resident launches are two-valued by the placement of the "
                                      "process's device memory (profiles/r05_hbm_bimodal.txt), constant inside a process -- the trace's duration is held against the traced process's own HIP-event period")
        rec = {"kernel": out["kernel_name"], "geometry": out["geometry"], "launch_mode": out["launch_mode_in_profiled_run"], "workload": workload, "head": hd, "kernel_sources_sha16": kernel_sources_sha16(),
               "hbm_bytes_per_launch": out.get("hbm_bytes_per_launch"), "hbm_read_bytes_per_launch": out.get("hbm_read_bytes_per_launch_corrected"),
               "hbm_write_bytes_per_launch": out.get("hbm_write_bytes_per_launch"), "l2_hit_rate": out.get("l2_hit_rate"),
               "half_steps_per_launch": hs, "body_us": body, "boundary_us": boundary, "period_us_in_kernel": in_kernel, "body_boundary_source": src,
               "burnin": ({k: burn[k] for k in ("body_us", "boundary_us", "period_us_in_kernel")} if burn and burn.get("geometry") == out["geometry"] else None),
               "period_us_unprofiled": per_cred or live, "period_us_unprofiled_burnin": per_burn,
               "rocprof_avg_duration_us": dur, "rocprof_inflated": inflated, "period_us_in_profiled_run": inprof, "two_valued_note": out.get("two_valued_note"),
               "duration_us": duration, "duration_source": dsrc,
               "source": f"profiles/{tag}_{cfg}_summary.json"}
        tpath = os.path.join(DST, f"traffic_{name}.json")
        if os.path.exists(tpath):                     # (the periods other boxes measured for this geometry stay on record: scripts/recompute_roofline.py)
            old = json.load(open(tpath))
            if "period_us_unprofiled_other_runs" in old and old.get("geometry") == rec.get("geometry"):
                rec["period_us_unprofiled_other_runs"] = old["period_us_unprofiled_other_runs"]
        json.dump(rec, open(tpath, "w"), indent=1)
        print(json.dumps({k: out.get(k) for k in ("config", "geometry", "launch_mode", "rocprof_inflated", "hbm_bytes_per_launch", "read_traffic_over_algorithmic_read", "l2_hit_rate", "fractions")}, indent=1))
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
            for name in (f"probe_{cfg}.txt", f"probe_light_{cfg}.txt"):
                p = os.path.join(SRC, name)
                if os.path.exists(p):
                    f.write(f"#### {name}\n" + open(p).read())


if __name__ == "__main__":
    main()
