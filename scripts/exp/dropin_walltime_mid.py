"""Wall time of the drop-in call kmc.emcee(...) on mid-size ensembles (the reference's users: 10^3-10^4 walkers of a few parameters),
by part, next to the device time of the generations alone.  Usage (GPU box): python scripts/exp/dropin_walltime_mid.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc

kmc.emcee(kmc.GaussianIso(), np.random.default_rng(0).standard_normal((256, 4)), niter=256 * 200, use_progress_meter=False, seed=1)      # warm the process
for nw, nd, G, nthin in ((4096, 4, 2000, 1), (4096, 4, 20000, 10), (16384, 4, 2000, 1), (4096, 32, 2000, 1), (1000, 8, 10000, 1)):
    th = np.random.default_rng(0).standard_normal((nw, nd))
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        thetas, acc, logd, _ = kmc.emcee(kmc.GaussianIso(), th, niter=nw * G, nthin=nthin, use_progress_meter=False, seed=5)
        best = min(best, time.perf_counter() - t)
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, nthin, 2.0, 5, store_chain=True, store_logp=True, chain_by_walker=True) as s:
        t = time.perf_counter(); s.set_positions(th); t_set = time.perf_counter() - t
        t = time.perf_counter(); s.run(G); s.sync(); t_run = time.perf_counter() - t
        dev = s.last_run_ms()
        t = time.perf_counter(); a, la = s.chain(by_walker=True); t_bw = time.perf_counter() - t
        how = s.describe().split(",")[0]
    print(f"kmc.emcee {nw} x {nd}, {G} generations, nthin {nthin}: {best * 1e3:7.2f} ms wall (best of 3), chain {thetas.nbytes / 1e6:.0f} MB | "
          f"device time of the generations {dev:6.2f} ms | set_positions {t_set * 1e3:5.2f} | run + sync {t_run * 1e3:6.2f} | chain by walker {t_bw * 1e3:6.2f} | {how}", flush=True)
