"""Whole benchmark jobs on the GPU and on the CPU oracle, compared at the end: final positions and acceptance counters bit for
bit, log-pdfs and moments to rounding.  C2 = the headline job (65 536 walkers x 32-dim Gaussian, 10^4 generations, bench.py's
exact inputs); C3 / C5 / C1 = bench.py's other_configs jobs.  Oracle time on the GPU box's CPU share: seconds to a minute.
Usage: python scripts/fulljob_parity.py [C2|C3|C5|C1 ...]        (default: all four)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import kissmcmc_jl_amd as kmc
import oracle


def jobs():
    rng = np.random.default_rng(bench.SEED)                       # the draws of bench.other_configs, in its order
    c1 = 0.5 + 0.1 * np.abs(rng.standard_normal((100, 1)))
    c3 = 0.1 * rng.standard_normal((16384, 64))
    c5 = rng.standard_normal((8192, 1024))
    return {
        "C2": (kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0], bench.theta0_c2(bench.NWALKERS_PER_GPU), 10000),
        "C3": (kmc.Rosenbrock(), oracle.ROSENBROCK, [1.0, 100.0, 20.0], c3, 10000),
        "C5": (kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0], c5, 2000),
        "C1": (kmc.Exponential(), oracle.EXPONENTIAL, [1.0], c1, 1000),
    }


def compare(name, job, cores=None):
    """One whole job on the default planner against the oracle's uninterrupted run: (ok, report line, describe)."""
    pdf, did, params, th, G = job
    cores = cores or bench.host_threads()
    nw, nd = th.shape
    nburn = G // 2
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, bench.SEED, moments=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        ms = s.last_run_ms()
        pos, nacc, logp = s.positions(), s.naccept(), s.logp()
        msum, msq, n = s.moments()
        how = s.describe()
    print(f"{name} GPU: {nw} x {nd}, {G} generations in {ms:.1f} ms = {nw * G / ms / 1e6:.3f}e9 walker-steps/s; {how}", flush=True)
    t0 = time.perf_counter()
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, 1, 2.0, bench.SEED, nthreads=cores), th, store_chain=False)
    print(f"{name} oracle: {time.perf_counter() - t0:.1f} s on {cores} threads", flush=True)
    same_pos = bool(np.array_equal(pos, ref["final_pos"]))
    same_acc = bool(np.array_equal(nacc, ref["naccept"]))
    dlogp = float(np.max(np.abs(logp - ref["final_logp"]) / np.maximum(1.0, np.abs(ref["final_logp"]))))
    dsum = float(np.max(np.abs(msum - ref["sum"]) / np.maximum(1e-9 * n, np.abs(ref["sum"]))))
    dsq = float(np.max(np.abs(msq - ref["sumsq"]) / np.abs(ref["sumsq"])))
    ok = same_pos and same_acc and n == ref["nmoment"] and dlogp < 1e-12 and dsq < 1e-11 and dsum < 1e-11
    line = (f"{name}: final positions bit-identical: {same_pos}; acceptance counters identical: {same_acc} "
            f"({int(nacc.sum())} accepted of {nw * (G - nburn)} counted proposals); max rel. log-pdf difference {dlogp:.2e}; "
            f"moments: nmoment {n} == {ref['nmoment']}, max rel. difference sum {dsum:.2e}, sumsq {dsq:.2e} -> {'OK' if ok else 'MISMATCH'}")
    print(line, flush=True)
    return ok, line, how


def main():
    names = sys.argv[1:] or ["C2", "C3", "C5", "C1"]
    J = jobs()
    sys.exit(1 if sum(not compare(name, J[name])[0] for name in names) else 0)


if __name__ == "__main__":
    main()
