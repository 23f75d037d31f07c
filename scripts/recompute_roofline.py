"""Recompute every fraction of a bench line's `roofline` objects from what is tracked under profiles/ -- the check VERDICT r03 asked a
reader to be able to make, hardened in round 6: for the headline config and every `other_configs` entry that carries a roofline,

    achieved      = (2 ndim + 1) * 8 B  x  walkers per launch  /  launch period            (algorithmic read)
    frac          = achieved / 8 TB/s  (period-based);     frac_of_measured_copy_rate = achieved / 6.29 TB/s
    duration_frac = the same bytes / duration_us / 8 TB/s                                  (duration_us, duration_source: profiles/traffic_<cfg>.json)
    body_frac     = the same bytes / body_us / 8 TB/s                                      (body_us: the light in-kernel probe, same record)
    traffic       = profiles/<tag>_<cfg>_summary.json: 2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 read correction)
    served_from   = state bytes against the 256 MiB Infinity Cache

and FAILS (exit status 1) when
  * a fraction of the line deviates from its recomputation by more than the tolerance (default 3 %);
  * the record attached to an entry was taken in another LAUNCH MODE than the line ran (`launch_mode`: table graph / updated graph / eager);
  * the record's kernel duration does not fit the line: duration_us x launches > the time that contains them (per phase where the line has the phases), or
    body_us > duration_us; a probe build whose in-kernel launch period is more than PROBE_TOL = 5 % off the production build's period of the same passes;
    a record whose production period is more than BOX_TOL = 4 % off the line's (another box);
  * the rocprofv3 mean duration exceeds the launch period and the record does not say so (`rocprof_inflated`) -- a kernel cannot take longer than the
    period that contains it; an inflated trace is allowed on record only next to another duration source.
No GPU needed.   python scripts/recompute_roofline.py [profiles/bench_r06a.json] [tolerance, default 0.03]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import launch_mode_of

PEAK, COPY, MALL = 8000.0, 6290.0, 256 * 2 ** 20
PROBE_TOL = 0.05      # what the light probe build may cost: its in-kernel launch period against the production build's period of the SAME passes (same box)
BOX_TOL = 0.04        # launch periods of one build on different boxes (round 6, five boxes: C2 credited 3.97-4.03 us, C3 credited 5.96-6.18)


def shape_of(name, line, entry):
    if name == "headline":
        return line["config"]["nwalkers_total"] // line["n_gpus"], line["config"]["ndim"]
    m = re.search(r"(\d+) walkers x (\d+)", entry.get("workload", "")) or re.search(r"(\d+) x (\d+)", entry.get("workload", ""))
    return int(m.group(1)), int(m.group(2))


def check(name, roof, nw, nd, tol, out, execution, credited_us=None, burnin_us=None):
    """-> (largest deviation of a held figure, [failures])"""
    worst, fails = 0.0, []

    def dev(label, mine, theirs, limit=tol):
        nonlocal worst
        if theirs in (None, 0) or mine is None:
            out.append(f"    {label:40s} -- (not on record)")
            return
        d = abs(mine / theirs - 1.0)
        worst = max(worst, d / limit * tol)                      # (normalised: a figure with its own limit counts against that limit)
        out.append(f"    {label:40s} recomputed {mine:14.6g}   line {theirs:14.6g}   dev {100 * d:5.2f} %" + (f"   <-- beyond {100 * limit:.0f} %" if d > limit else ""))

    def fail(msg):
        fails.append(f"{name}: {msg}")
        out.append("    FAIL: " + msg)

    hs = roof.get("half_steps_per_launch") or (2 if "generation_" in (roof.get("geometry") or "") else 1)      # one launch per generation: two half-steps' walkers and time
    alg = hs * (nw // 2) * (2 * nd + 1) * 8
    period = roof["avg_launch_us"]
    credited_us = credited_us * hs if credited_us else None     # (other_configs quote microseconds per half-step)
    burnin_us = burnin_us * hs if burnin_us else None
    out.append(f"  {name}: {nw} x {nd}, kernel {roof['kernel']}, {roof['geometry']}, launch mode {launch_mode_of(execution)}")
    dev("algorithmic read bytes / launch", alg, roof["algorithmic_read_bytes_per_launch"])
    dev("achieved GB/s (line's HIP events)", alg / (period * 1e-6) / 1e9, roof["achieved"])
    dev("frac of 8 TB/s (period-based)", alg / (period * 1e-6) / 1e9 / PEAK, roof["frac"])
    dev("frac of the 6.29 TB/s copy rate", alg / (period * 1e-6) / 1e9 / COPY, roof["frac_of_measured_copy_rate"])
    served = "infinity_cache" if roof["state_bytes"] <= MALL else "hbm"
    out.append(f"    served_from: state {roof['state_bytes'] / 2 ** 20:.0f} MiB -> {served}" + ("" if served == roof["served_from"] else "   <-- the line says " + roof["served_from"]))
    if served != roof["served_from"]:
        fail(f"served_from is {served}, the line says {roof['served_from']}")
    rec_name = re.search(r"profiles/(traffic_\w+\.json)", (roof.get("profile_record") or {}).get("record") or roof.get("traffic_source", ""))    # (round-4 lines: in the prose)
    rec = json.load(open(os.path.join(ROOT, "profiles", rec_name.group(1)))) if rec_name and roof.get("traffic") else None
    if rec is None:
        out.append("    (no profile record attached to this entry)")
        return worst, fails
    out.append(f"    record {rec_name.group(1)} (head {rec['head']}): geometry {'matches' if rec['geometry'] == roof['geometry'] else 'DIFFERS'}, "
               f"launch mode {rec.get('launch_mode', '(not on record)')}")
    if rec["geometry"] != roof["geometry"]:
        fail("the record's kernel geometry DIFFERS from the line's")
    if "launch_mode" in rec and rec["launch_mode"] != launch_mode_of(execution):
        fail(f"the record was taken in launch mode {rec['launch_mode']}, the line ran {launch_mode_of(execution)}: not evidence for this line")
    # the record's own unprofiled period against the line's (per phase where both have it; the HBM-resident launches are two-valued by process: the nearest on record)
    if credited_us and rec.get("period_us_unprofiled"):
        dev("record's period (credited) vs line's", rec["period_us_unprofiled"], credited_us, BOX_TOL)
        if burnin_us and rec.get("period_us_unprofiled_burnin"):
            dev("record's period (burn-in) vs line's", rec["period_us_unprofiled_burnin"], burnin_us, BOX_TOL)
    else:
        periods = [rec["period_us_unprofiled"]] + list(rec.get("period_us_unprofiled_other_runs", {}).get("values", []))
        nearest = min(periods, key=lambda p: abs(p - period))
        if nearest != rec["period_us_unprofiled"]:
            out.append(f"    (the record's own period is {rec['period_us_unprofiled']:.2f} us; of the {len(periods) - 1} other runs of this geometry on record the nearest is {nearest:.2f} us)")
        dev("frac from the record's period", alg / (nearest * 1e-6) / 1e9 / PEAK, roof["frac"])
    # the three fractions side by side, and the containment  body <= duration <= period  they must satisfy
    dur, body = rec.get("duration_us"), rec.get("body_us")
    probe_source = "probe" in (rec.get("duration_source") or "")
    held = credited_us or period                                  # the time one launch of the record's kind (moments credited) has in the line
    out.append(f"    fractions of 8 TB/s: period-based {roof['frac']:.3f}   duration-based {roof.get('duration_frac') or float('nan'):.3f} ({rec.get('duration_source', 'no duration on record')})   "
               f"body-based {roof.get('body_frac') or float('nan'):.3f}")
    if "rocprof_inflated" in rec:                                 # (records since round 6)
        if rec["rocprof_avg_duration_us"] > held * (1.0 + tol) and not rec["rocprof_inflated"] and not (served == "hbm" and rec.get("two_valued_note")):
            fail(f"rocprofv3's mean duration {rec['rocprof_avg_duration_us']:.2f} us exceeds the launch period {held:.2f} us and the record does not say so")
        if rec["rocprof_inflated"]:
            out.append(f"    rocprofv3's mean duration {rec['rocprof_avg_duration_us']:.2f} us > the period {held:.2f} us: on record as INFLATED BY THE TOOL, not used; duration source: {rec.get('duration_source')}")
            if dur is None:
                fail("the rocprofv3 duration is inflated and the record has no other duration source")
        if dur is not None:
            limit = PROBE_TOL if probe_source else tol
            own = rec.get("period_us_in_profiled_run")
            if dur <= held * (1.0 + limit):
                out.append(f"    duration {dur:.2f} us <= {held:.2f} us per launch (+ {100 * limit:.0f} %): fits")
            elif served == "hbm" and rec.get("two_valued_note") and own and dur <= own * (1.0 + tol):
                # (the HBM-resident launches are two-valued by process: the traced process sat in the other mode than this line's; its duration fits its own period)
                out.append(f"    duration {dur:.2f} us > this line's {held:.2f} us but <= the traced process's own HIP-event period {own:.2f} us: two-valued by process, on record as such")
            else:
                fail(f"duration_us {dur:.2f} x launches does not fit the time that contains them ({held:.2f} us per launch, + {100 * limit:.0f} %)")
            dev("duration_frac", alg / (dur * 1e-6) / 1e9 / PEAK, roof.get("duration_frac"))
            if body is not None and body > dur * (1.0 + tol):
                fail(f"body_us {body:.2f} exceeds duration_us {dur:.2f}")
    if body is not None and rec.get("boundary_us") is not None:
        if rec.get("period_us_in_kernel") is not None:            # light probe: body + boundary of the probe build against the production build's period of the same passes
            dev("probe build's period vs production's", rec["period_us_in_kernel"], rec["period_us_unprofiled"], PROBE_TOL)
            if rec.get("burnin") and rec.get("period_us_unprofiled_burnin"):
                dev("... in burn-in", rec["burnin"]["period_us_in_kernel"], rec["period_us_unprofiled_burnin"], PROBE_TOL)
        else:
            out.append(f"    body + boundary ({'trace duration, period - duration' if 'kernel trace' in (rec.get('body_boundary_source') or '') else 'full probe build, a round-5 record'})  "
                       f"{body:.2f} + {rec['boundary_us']:.2f} = {body + rec['boundary_us']:.2f} us against the line's {period:.2f} us")
        dev("body_frac", alg / (body * 1e-6) / 1e9 / PEAK, roof["body_frac"])
    summ = json.load(open(os.path.join(ROOT, rec["source"])))
    if "pmc_per_launch" in summ:
        pmc = summ["pmc_per_launch"]
        fetch, write = pmc.get("FETCH_SIZE"), pmc.get("WRITE_SIZE")          # KiB per dispatch; the summaries use the second half of the dispatches
        if fetch and write:
            dev("traffic = (2 x FETCH + WRITE) KiB", (2 * fetch["second_half_mean"] + write["second_half_mean"]) * 1024, roof["traffic"])
    else:
        dev("traffic (summary)", summ.get("hbm_bytes_per_launch"), roof["traffic"])
    dev("read traffic / algorithmic read", rec["hbm_read_bytes_per_launch"] / alg, rec["hbm_read_bytes_per_launch"] / roof["algorithmic_read_bytes_per_launch"])
    return worst, fails


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "bench_r06a.json")
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
    text = open(path).read()
    try:
        lines = [json.loads(text)]               # bench_detail.json itself (one indented record)
    except ValueError:
        lines = [json.loads(l) for l in text.splitlines() if l.lstrip().startswith("{")]
    detail = [l["bench_detail"] for l in lines if "bench_detail" in l]      # since round 5 the full record is a line of its own (or bench_detail.json itself)
    line = detail[-1] if detail else lines[-1]
    out = [f"{os.path.relpath(path, ROOT)}: value {line['value']:.4g} {line['unit']}" + (f" (median of {line['repetitions']}: {line['value_min']:.4g} .. {line['value_max']:.4g})" if "repetitions" in line else "") +
           f", {line['ms_per_step']:.3f} ms per step of {line['config']['gens_per_step']} generations"]
    launches = line["steps"] * line["config"]["gens_per_step"] * 2
    out.append(f"  ms_per_step / launches per step = {line['ms_per_step'] * 1e3 / (launches / line['steps']):.4f} us per launch (the line's avg_launch_us: {line['roofline']['avg_launch_us']:.4f})")
    nw, nd = shape_of("headline", line, None)
    roof = line["roofline"]
    worst, fails = check("headline (C2)", roof, nw, nd, tol, out, line["config"]["execution"],
                         roof.get("avg_launch_us_credited"), roof.get("avg_launch_us_burnin"))
    for key, entry in (line.get("other_configs") or {}).items():
        if isinstance(entry, dict) and isinstance(entry.get("roofline"), dict):
            nw, nd = shape_of(key, line, entry)
            w, f = check(key, entry["roofline"], nw, nd, tol, out, entry.get("execution"), entry.get("us_per_half_step_after_burnin"), entry.get("us_per_half_step_burnin"))
            worst, fails = max(worst, w), fails + f
    out.append(f"largest deviation: {100 * worst:.2f} % of a 3 % tolerance (normalised: probe-build periods are held to {100 * PROBE_TOL:.0f} %, periods of another box to {100 * BOX_TOL:.0f} %)")
    out += [f"FAILED: {f}" for f in fails]
    print("\n".join(out))
    return 0 if worst <= tol and not fails else 1


if __name__ == "__main__":
    sys.exit(main())
