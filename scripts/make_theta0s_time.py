import sys, time
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
pdf = kmc.Exponential(1.0)
kmc.make_theta0s(0.5, 0.1, pdf, 100, rng=3)
for nw, nd in ((100, 1), (100, 2), (1000, 4), (65536, 32)):
    th0 = 0.5 if nd == 1 else np.full(nd, 0.5)
    p = pdf if nd == 1 else kmc.GaussianIso(0.0, 1.0)
    t0 = time.perf_counter(); th = kmc.make_theta0s(th0, 0.1, p, nw, rng=3); t1 = time.perf_counter()
    print(f"make_theta0s {nw} x {nd}: {1e3 * (t1 - t0):.2f} ms", flush=True)
cd = kmc.CDensity("return x[0] >= 0 ? -x[0] : -INFINITY;")
t0 = time.perf_counter(); th = kmc.make_theta0s(0.5, 0.1, cd, 100, rng=3); t1 = time.perf_counter()
print(f"make_theta0s CDensity 100 x 1: {1e3 * (t1 - t0):.2f} ms")
t0 = time.perf_counter(); th = kmc.make_theta0s(0.5, 0.1, cd, 100, rng=3); t1 = time.perf_counter()
print(f"make_theta0s CDensity 100 x 1 (again): {1e3 * (t1 - t0):.2f} ms")
t0 = time.perf_counter(); th = kmc.make_theta0s(0.5, 0.1, lambda x: -x if x >= 0 else -np.inf, 100, rng=3); t1 = time.perf_counter()
print(f"make_theta0s python callable 100 x 1: {1e3 * (t1 - t0):.2f} ms")
