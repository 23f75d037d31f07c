"""Read a multi-GPU record of bench.py -- SCALE_rNN.json as the driver writes it, a file of bench.py's JSON lines, or several files --
and say, per N, what ran and what bounded it: value, efficiency against N = 1, the rung that supplied `value`, every rung of the ladder
that failed or timed out, whether the timed run equalled the unsharded run, and the fabric arithmetic (bytes per link per launch, the link's
MEASURED rate, the link-bound time against the measured launch period, the exact rule's projected speed-up next to the measured one).  No GPU needed.  python scripts/scale_report.py SCALE_r04.json [more.json ...]"""
import json
import sys


def lines_of(obj):
    """Every dict that looks like a bench.py result line, wherever the driver put it."""
    if isinstance(obj, dict):
        if obj.get("metric") == "walker-steps/sec" and "n_gpus" in obj:
            yield obj
        for v in obj.values():
            yield from lines_of(v)
    elif isinstance(obj, list):
        for v in obj:
            yield from lines_of(v)
    elif isinstance(obj, str) and obj.lstrip().startswith('{"metric"'):
        try:
            yield from lines_of(json.loads(obj))
        except ValueError:
            pass


def main():
    found = {}
    for path in sys.argv[1:]:
        text = open(path).read()
        try:
            objs = [json.loads(text)]
        except ValueError:
            objs = [json.loads(l) for l in text.splitlines() if l.lstrip().startswith("{")]
        for o in objs:
            for line in lines_of(o):
                found[line["n_gpus"]] = line
    if not found:
        print("no bench.py result line found (a skipped record?)")
        return 1
    base = found.get(1, {}).get("value")
    for n in sorted(found):
        r = found[n]
        eff = f"{r['value'] / (n * base):.2f} of linear" if base else "no N = 1 line"
        print(f"N = {n}: {r['value']:.3e} walker-steps/s ({eff}); {r['ms_per_step']:.2f} ms per step; {r['config']['parallelism']}")
        roof = r.get("roofline", {})
        print(f"   launch period {roof.get('avg_launch_us', float('nan')):.2f} us, {roof.get('launches')} launches, kernel {roof.get('kernel')}")
        if n > 1:
            print(f"   value from: {r.get('value_from')}; timed run == unsharded run: {r.get('check', {}).get('timed_run_equals_unsharded_run')}")
            for rung in r.get("ladder", []):
                if not rung.get("ok", True):
                    print(f"   rung NOT ok: {rung['rung']} after {rung['s']} s" + (" (timed out)" if rung.get("timed_out") else ""))
            fab = r.get("fabric", {})
            if fab:
                if "link_bound_us" in fab:            # since round 6: the link's rate is measured by the line's own `link-probe` rung
                    print(f"   fabric: {fab['bytes_per_link_per_launch'] / 1e6:.2f} MB per link per launch at {fab.get('link_gather_GBs') or 77.0:.1f} GB/s ({fab.get('link_rate_source')}; "
                          f"runtime copy {fab.get('link_copy_GBs') or float('nan'):.1f}, same pattern on local memory {fab.get('local_gather_GBs') or float('nan'):.0f}) -> >= {fab['link_bound_us']:.1f} us "
                          f"against the measured {roof.get('avg_launch_us', float('nan')):.1f} us; one GPU alone {fab.get('single_gpu_us_per_launch') or float('nan'):.2f} us per launch")
                    print(f"   exact partner rule (src/samplers.jl:250): projected speed-up {fab.get('projected_exact_speedup') or float('nan'):.2f} x, measured {fab.get('measured_speedup') or float('nan'):.2f} x; "
                          f">= 6 x expected: {fab.get('ge6x_expected_under_exact_rule')}")
                else:
                    print(f"   fabric: {fab['bytes_per_link_per_launch'] / 1e6:.2f} MB per link per launch -> >= {fab['link_bound_us_at_77GBs']:.1f} us at an ASSUMED 77 GB/s "
                          f"against the measured {roof.get('avg_launch_us', float('nan')):.1f} us")
            col = r.get("collective", {})
            print(f"   collective: {col.get('backend')} world {col.get('world_size')} (seen by all-reduce: {col.get('ranks_seen_by_all_reduce')}), RCCL {col.get('rccl_version')}, "
                  f"launcher: {col.get('launcher')}, rank env {col.get('rank_env')}")
            for key in ("dealt_mode", "allgather_mode"):
                x = r.get(key)
                if x:
                    print(f"   {key}: " + (f"ERROR {x['error']}" if "error" in x else f"{x['value']:.3e} walker-steps/s" + (f" ({x['value'] / (n * base):.2f} of linear)" if base else "")))
            if r.get("extras_timed_out"):
                print(f"   extras timed out: {r['extras_timed_out']}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
