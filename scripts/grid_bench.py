"""Throughput grid (exact mode, moments on) over ensemble size x ndim, isotropic Gaussian; writes
profiles/<tag>_grid.json and a markdown table.  Usage: python3 scripts/grid_bench.py r01c"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kissmcmc_jl_amd as kmc

tag = sys.argv[1] if len(sys.argv) > 1 else "r01c"
rows = []
for nw in (128, 1024, 8192, 65536, 524288):
    for nd in (2, 8, 32, 128, 1024):
        if nw < nd + 2 or nw * nd * 8 > 6 * 2 ** 30:
            continue
        th = np.random.default_rng(0).standard_normal((nw, nd))
        G = max(8, min(2048, int(2e9 / (nw * max(nd, 16)))))
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, 10 ** 9, 0, 1, 2.0, 1, moments=True) as s:
            s.set_positions(th)
            s.run(max(4, G // 4))
            s.sync()
            ts = []
            for _ in range(3):
                s.run(G)
                s.sync()
                ts.append(s.last_run_ms())
            t = min(ts)
            acc = float(s.naccept().mean() / s.generation)
        steps = nw * G / (t * 1e-3)
        r = dict(nwalkers=nw, ndim=nd, generations=G, us_per_generation=t / G * 1e3, walker_steps_per_s=steps,
                 algorithmic_read_GBs=steps * (2 * nd + 1) * 8 / 1e9, accept=acc)
        rows.append(r)
        print(r, flush=True)
json.dump(rows, open(os.path.join(ROOT, "profiles", f"{tag}_grid.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", f"{tag}_grid.md"), "w") as f:
    f.write("| nwalkers | ndim | µs / generation | walker-steps/s | algorithmic read GB/s | accept |\n|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r['nwalkers']} | {r['ndim']} | {r['us_per_generation']:.2f} | {r['walker_steps_per_s']:.3e} | "
                f"{r['algorithmic_read_GBs']:.0f} | {r['accept']:.3f} |\n")
