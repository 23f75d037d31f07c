"""Wall time of ONE Metropolis chain (the reference's metropolis(pdf, sample_ppdf, theta0; niter)) and of many chains at once."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc

pdf = kmc.GaussianIso(-5.0, 3.0)
step = kmc.GaussianStep(5.0)
for niter in (10 ** 5, 10 ** 5, 10 ** 6, 10 ** 7):
    t0 = time.perf_counter()
    th, acc, logd, _ = kmc.metropolis(pdf, step, 0.0, niter=niter, seed=3, use_progress_meter=False)
    t1 = time.perf_counter()
    print(f"one chain, niter = {niter:>9}: {1e3 * (t1 - t0):9.2f} ms -> {niter / (t1 - t0):.3e} steps/s wall; mean {np.mean(th):.3f} std {np.std(th):.3f} accept {acc:.3f}", flush=True)
for nthin in (10, 100):
    niter = 10 ** 7
    t0 = time.perf_counter()
    th, acc, logd, _ = kmc.metropolis(pdf, step, 0.0, niter=niter, nthin=nthin, seed=3, use_progress_meter=False)
    t1 = time.perf_counter()
    print(f"one chain, niter = {niter:>9}, nthin = {nthin}: {1e3 * (t1 - t0):9.2f} ms -> {niter / (t1 - t0):.3e} steps/s wall", flush=True)
for nchains in (64, 1024, 4096, 16384, 65536):
    niter = 10 ** 4
    t0 = time.perf_counter()
    th, acc, logd, _ = kmc.metropolis_chains(pdf, step, np.zeros(nchains), niter=niter, seed=3)
    t1 = time.perf_counter()
    print(f"{nchains} chains x {niter} steps: {1e3 * (t1 - t0):9.2f} ms -> {nchains * niter / (t1 - t0):.3e} chain-steps/s wall", flush=True)
