"""The README call with the density as a plain Python callable (the host route: one round trip per half-step)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc

theta0s = kmc.make_theta0s(0.5, 0.1, kmc.Exponential(1.0), 100, rng=3)
def logpdf_exp(x):
    return -x if x >= 0 else -np.inf
vec = kmc.HostLogPdf(lambda X: np.where(X[:, 0] >= 0, -X[:, 0], -np.inf), vectorized=True)
for name, pdf in (("scalar callable", logpdf_exp), ("vectorized HostLogPdf", vec), ("device Exponential", kmc.Exponential(1.0))):
    for rep in range(2):
        t0 = time.perf_counter()
        thetas, acc, logd, _ = kmc.emcee(pdf, theta0s, niter=10 ** 5, seed=7, use_progress_meter=False)
        t1 = time.perf_counter()
    print(f"{name:24s}: {1e3 * (t1 - t0):8.2f} ms for niter = 10^5 (2000 half-steps) -> {1e6 * (t1 - t0) / 2000:.1f} us per half-step; mean {np.mean(thetas):.3f}", flush=True)
