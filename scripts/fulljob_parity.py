"""The whole headline job (C2: 65 536 walkers x 32-dim Gaussian, 10^4 generations, bench.py's exact inputs) on the GPU and
on the CPU oracle, compared bit for bit at the end: positions, acceptance counters, moments.  ~1 minute of oracle time on
the GPU box's host cores.  Usage: python scripts/fulljob_parity.py [generations]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import kissmcmc_jl_amd as kmc
import oracle

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nw, nd, nburn = bench.NWALKERS_PER_GPU, bench.NDIM, G // 2
th = bench.theta0_c2(nw)
with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, bench.SEED, moments=True) as s:
    s.set_positions(th)
    s.run(G)
    s.sync()
    ms = s.last_run_ms()
    pos, nacc, logp = s.positions(), s.naccept(), s.logp()
    msum, msq, n = s.moments()
    how = s.describe()
print(f"GPU: {G} generations in {ms:.1f} ms = {nw * G / ms / 1e6:.2f}e9 walker-steps/s; {how}", flush=True)
t0 = time.perf_counter()
cores = min(os.cpu_count() or 1, 64)
ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 1, 2.0, bench.SEED, nthreads=cores), th,
                   store_chain=False)
print(f"oracle: {time.perf_counter() - t0:.1f} s on {cores} threads", flush=True)
same_pos = bool(np.array_equal(pos, ref["final_pos"]))
same_acc = bool(np.array_equal(nacc, ref["naccept"]))
dlogp = float(np.max(np.abs(logp - ref["final_logp"]) / np.maximum(1.0, np.abs(ref["final_logp"]))))
dsum = float(np.max(np.abs(msum - ref["sum"]) / np.maximum(1e-9, np.abs(ref["sum"]))))
dsq = float(np.max(np.abs(msq - ref["sumsq"]) / np.abs(ref["sumsq"])))
print(f"final positions bit-identical: {same_pos}; acceptance counters identical: {same_acc} "
      f"({int(nacc.sum())} accepted of {nw * (G - nburn)} counted proposals); max rel. log-pdf difference {dlogp:.2e}; "
      f"moments: nmoment {n} == {ref['nmoment']}, max rel. difference sum {dsum:.2e}, sumsq {dsq:.2e}")
sys.exit(0 if (same_pos and same_acc and n == ref["nmoment"] and dlogp < 1e-12 and dsq < 1e-11) else 1)
