"""Throughput of the many-chain Metropolis kernel (chain-steps/s, HIP events around the sampling launches only)
next to the CPU oracle on this box's cores.  Usage: python scripts/metropolis_bench.py [niter]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd.metropolis import run_chains

CASES = [
    ("normal(-5,3) 1-D, step 9", lambda: kmc.GaussianIso(-5.0, 3.0), 1, 9.0, -4.0),
    ("rosenbrock 2-D, step 0.5", lambda: kmc.Rosenbrock(), 2, 0.5, 0.0),
    ("gaussian 8-D, step 0.5", lambda: kmc.GaussianIso(), 8, 0.5, 0.0),
    ("gaussian 16-D, step 0.4", lambda: kmc.GaussianIso(), 16, 0.4, 0.0),
    ("gaussian 32-D, step 0.3", lambda: kmc.GaussianIso(), 32, 0.3, 0.0),
    ("gaussian 64-D, step 0.2 (chain in memory)", lambda: kmc.GaussianIso(), 64, 0.2, 0.0),
]


def main():
    niter = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    for name, mk, nd, step, start in CASES:
        for nchains in (65536, 1048576):
            th = np.full((nchains, nd), start)
            pdf = mk()
            run_chains(pdf, kmc.GaussianStep(step), th, 64, 32, 1, 1, store_chain=False, store_logp=False, moments=True)   # warm-up
            r = run_chains(pdf, kmc.GaussianStep(step), th, niter, niter // 2, 1, 1, store_chain=False, store_logp=False, moments=True)
            rate = nchains * niter / (r["device_ms"] * 1e-3)
            print(f"{name:44s} {nchains:8d} chains x {niter} steps: {r['device_ms']:8.2f} ms  {rate / 1e9:7.2f} G chain-steps/s  "
                  f"accept {r['accept_ratio'].mean():.3f}", flush=True)
    try:
        import oracle
        oracle.build()
        cores = min(os.cpu_count() or 1, 64)
        th = np.full((4096, 1), -4.0)
        t0 = time.perf_counter()
        oracle.metropolis(oracle.GAUSSIAN_ISO, [-5.0, 3.0], th, 9.0, 2000, 1000, 1, 1, nthreads=cores, store_chain=False)
        dt = time.perf_counter() - t0
        print(f"CPU oracle, {cores} threads, normal 1-D: {4096 * 2000 / dt / 1e6:.1f} M chain-steps/s")
    except Exception as e:  # noqa: BLE001
        print("oracle not timed:", e)


if __name__ == "__main__":
    main()
