"""Scratch timing of the half-step kernel at the BASELINE configs for several launch geometries.
Usage: python scripts/quick_bench.py [config] [plans...]   (plans like 16,1,4)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc

CONFIGS = {
    "C2": (kmc.GaussianIso(), 65536, 32, 2048),
    "C3": (kmc.Rosenbrock(), 16384, 64, 2048),
    "C5": (kmc.GaussianIso(), 8192, 1024, 512),
    "C1": (kmc.Exponential(), 100, 1, 1024),
}


def run(cfgname, plan, moments=True, graph=True):
    pdf, nw, nd, G = CONFIGS[cfgname]
    if plan:
        os.environ["KMC_PLAN"] = plan
    else:
        os.environ.pop("KMC_PLAN", None)
    rng = np.random.default_rng(0)
    th = rng.standard_normal((nw, nd)) if cfgname != "C1" else 0.5 + 0.1 * np.abs(rng.standard_normal((nw, nd)))
    if cfgname == "C3":
        th *= 0.1
    with kmc.Sampler(pdf, nw, nd, 10 ** 9, 0, 1, 2.0, 7, moments=moments, use_graph=graph) as s:
        s.set_positions(th)
        s.run(256)
        s.sync()
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            s.run(G)
            s.sync()
            wall = time.perf_counter() - t0
            ms = s.last_run_ms()
            best = min(best, ms)
        acc = s.naccept().mean() / s.generation
        steps = nw * G / (best * 1e-3)
        br = (2 * nd + 1) * 8
        print(f"{cfgname} plan={plan or 'default':10s} moments={int(moments)} graph={int(graph)} "
              f"{best / (2 * G) * 1e3:8.2f} us/half-step  {steps / 1e9:7.3f} Gsteps/s  "
              f"read {steps * br / 1e12:6.3f} TB/s  wall {wall * 1e3:.1f} ms acc={acc:.3f}", flush=True)


if __name__ == "__main__":
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
    plans = sys.argv[2:] or [""]
    for p in plans:
        for mom in (True, False):
            run(cfgname, p, moments=mom)
    run(cfgname, plans[0], moments=True, graph=False)
