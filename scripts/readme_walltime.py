"""Wall time of the reference's README call through the drop-in surface (make_theta0s -> emcee -> squash_walkers), by niter."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc

pdf = kmc.Exponential(1.0)
theta0s = kmc.make_theta0s(0.5, 0.1, pdf, 100, rng=3)
for niter in (10 ** 5, 10 ** 5, 10 ** 5, 10 ** 6, 10 ** 7, 10 ** 8):
    t0 = time.perf_counter()
    thetas, acc, logd, _ = kmc.emcee(pdf, theta0s, niter=niter, seed=7, use_progress_meter=False)
    t1 = time.perf_counter()
    th = kmc.squash_walkers(thetas, acc, logd, verbose=False)[0]
    t2 = time.perf_counter()
    print(f"niter = {niter:>9}: emcee {1e3 * (t1 - t0):8.2f} ms  squash {1e3 * (t2 - t1):7.2f} ms  "
          f"-> {niter / (t1 - t0):.3e} evaluations/s wall; samples {np.asarray(th).shape}, mean {np.mean(th):.4f}, accept {np.mean(acc):.3f}", flush=True)
