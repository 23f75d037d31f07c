"""Recompute every fraction of a bench line's `roofline` objects from what is tracked under profiles/ -- the check VERDICT r03 asked a
reader to be able to make: for the headline config and every `other_configs` entry that carries a roofline,

    achieved      = (2 ndim + 1) * 8 B  x  walkers per launch  /  launch period            (algorithmic read)
    frac          = achieved / 8 TB/s;     frac_of_measured_copy_rate = achieved / 6.29 TB/s
    body_frac     = the same bytes / body_us / 8 TB/s                                      (body_us: profiles/traffic_<cfg>.json)
    traffic       = profiles/<tag>_<cfg>_summary.json: 2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 read correction)
    served_from   = state bytes against the 256 MiB Infinity Cache

with the launch period taken three ways -- the line's own HIP events, the period the profile passes measured unprofiled
(`period_us_unprofiled` of the record), and body_us + boundary_us of the probe build -- and prints the deviation of each from the line.
No GPU needed.   python scripts/recompute_roofline.py [profiles/bench_r04i.json] [tolerance, default 0.03]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK, COPY, MALL = 8000.0, 6290.0, 256 * 2 ** 20


def shape_of(name, line, entry):
    if name == "headline":
        return line["config"]["nwalkers_total"] // line["n_gpus"], line["config"]["ndim"]
    m = re.search(r"(\d+) walkers x (\d+)", entry.get("workload", "")) or re.search(r"(\d+) x (\d+)", entry.get("workload", ""))
    return int(m.group(1)), int(m.group(2))


def check(name, roof, nw, nd, tol, out, after_burnin_us=None):
    worst = 0.0

    def dev(label, mine, theirs):
        nonlocal worst
        if theirs in (None, 0) or mine is None:
            out.append(f"    {label:34s} -- (not on record)")
            return
        d = abs(mine / theirs - 1.0)
        worst = max(worst, d)
        out.append(f"    {label:34s} recomputed {mine:14.6g}   line {theirs:14.6g}   dev {100 * d:5.2f} %" + ("   <-- beyond tolerance" if d > tol else ""))

    hs = roof.get("half_steps_per_launch") or (2 if "generation_" in (roof.get("geometry") or "") else 1)      # one launch per generation: two half-steps' walkers and time
    alg = hs * (nw // 2) * (2 * nd + 1) * 8
    if after_burnin_us:
        after_burnin_us *= hs                                                                                  # (the line quotes microseconds per half-step)
    out.append(f"  {name}: {nw} x {nd}, kernel {roof['kernel']}, {roof['geometry']}")
    dev("algorithmic read bytes / launch", alg, roof["algorithmic_read_bytes_per_launch"])
    dev("achieved GB/s (line's HIP events)", alg / (roof["avg_launch_us"] * 1e-6) / 1e9, roof["achieved"])
    dev("frac of 8 TB/s", alg / (roof["avg_launch_us"] * 1e-6) / 1e9 / PEAK, roof["frac"])
    dev("frac of the 6.29 TB/s copy rate", alg / (roof["avg_launch_us"] * 1e-6) / 1e9 / COPY, roof["frac_of_measured_copy_rate"])
    served = "infinity_cache" if roof["state_bytes"] <= MALL else "hbm"
    out.append(f"    served_from: state {roof['state_bytes'] / 2 ** 20:.0f} MiB -> {served}" + ("" if served == roof["served_from"] else "   <-- the line says " + roof["served_from"]))
    rec_name = re.search(r"profiles/(traffic_\w+\.json)", (roof.get("profile_record") or {}).get("record") or roof.get("traffic_source", ""))    # (round-4 lines: in the prose)
    rec = json.load(open(os.path.join(ROOT, "profiles", rec_name.group(1)))) if rec_name and roof.get("traffic") else None
    if rec is None:
        out.append("    (no profile record attached to this entry)")
        return worst
    out.append(f"    record {rec_name.group(1)} (head {rec['head']}): geometry {'matches' if rec['geometry'] == roof['geometry'] else 'DIFFERS'}")
    if after_burnin_us:       # the records are taken from launches that credit moments; a job's first half (burn-in) credits none and is cheaper
        dev("record's period vs after burn-in", rec["period_us_unprofiled"], after_burnin_us)
    else:
        # the record's own unprofiled period -- or, where the record lists the periods other boxes measured for this geometry (the HBM-resident
        # launches are bimodal from box to box), the one nearest to this line's
        periods = [rec["period_us_unprofiled"]] + list(rec.get("period_us_unprofiled_other_runs", {}).get("values", []))
        nearest = min(periods, key=lambda p: abs(p - roof["avg_launch_us"]))
        if nearest != rec["period_us_unprofiled"]:
            out.append(f"    (the record's own period is {rec['period_us_unprofiled']:.2f} us; of the {len(periods) - 1} other runs of this geometry on record the nearest is {nearest:.2f} us)")
        dev("frac from the record's period", alg / (nearest * 1e-6) / 1e9 / PEAK, roof["frac"])
    if rec.get("body_us") is not None and rec.get("boundary_us") is not None:
        out.append(f"    body + boundary in the probe build   {rec['body_us']:.2f} + {rec['boundary_us']:.2f} = {rec['body_us'] + rec['boundary_us']:.2f} us against the line's {roof['avg_launch_us']:.2f} us "
                   "(the in-kernel stamps cost time in the short kernels: informative, not held to the tolerance)")
        dev("body_frac", alg / (rec["body_us"] * 1e-6) / 1e9 / PEAK, roof["body_frac"])
    summ = json.load(open(os.path.join(ROOT, rec["source"])))
    if "pmc_per_launch" in summ:
        pmc = summ["pmc_per_launch"]
        fetch, write = pmc.get("FETCH_SIZE"), pmc.get("WRITE_SIZE")          # KiB per dispatch; the summaries use the second half of the dispatches
        if fetch and write:
            dev("traffic = (2 x FETCH + WRITE) KiB", (2 * fetch["second_half_mean"] + write["second_half_mean"]) * 1024, roof["traffic"])
    else:
        dev("traffic (summary)", summ.get("hbm_bytes_per_launch"), roof["traffic"])
    dev("read traffic / algorithmic read", rec["hbm_read_bytes_per_launch"] / alg, rec["hbm_read_bytes_per_launch"] / roof["algorithmic_read_bytes_per_launch"])
    return worst


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "bench_r04i.json")
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
    text = open(path).read()
    try:
        lines = [json.loads(text)]               # bench_detail.json itself (one indented record)
    except ValueError:
        lines = [json.loads(l) for l in text.splitlines() if l.lstrip().startswith("{")]
    detail = [l["bench_detail"] for l in lines if "bench_detail" in l]      # since round 5 the full record is a line of its own (or bench_detail.json itself)
    line = detail[-1] if detail else lines[-1]
    out = [f"{os.path.relpath(path, ROOT)}: value {line['value']:.4g} {line['unit']}, {line['ms_per_step']:.3f} ms per step of {line['config']['gens_per_step']} generations"]
    launches = line["steps"] * line["config"]["gens_per_step"] * 2
    out.append(f"  ms_per_step / launches per step = {line['ms_per_step'] * 1e3 / (launches / line['steps']):.4f} us per launch (the line's avg_launch_us: {line['roofline']['avg_launch_us']:.4f})")
    worst = 0.0
    nw, nd = shape_of("headline", line, None)
    worst = max(worst, check("headline (C2)", line["roofline"], nw, nd, tol, out))
    for key, entry in (line.get("other_configs") or {}).items():
        if isinstance(entry, dict) and isinstance(entry.get("roofline"), dict):
            nw, nd = shape_of(key, line, entry)
            worst = max(worst, check(key, entry["roofline"], nw, nd, tol, out, entry.get("us_per_half_step_after_burnin")))
    out.append(f"largest deviation: {100 * worst:.2f} % (tolerance {100 * tol:.0f} %)")
    print("\n".join(out))
    return 0 if worst <= tol else 1


if __name__ == "__main__":
    sys.exit(main())
