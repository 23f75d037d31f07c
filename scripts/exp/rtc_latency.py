"""First-use latency of runtime-compiled densities (hiprtc) in a fresh process: creating the density (syntax check), the
first sampler of a geometry (kernel instantiation), a second sampler of the same geometry (in-process cache)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc
th = np.random.default_rng(0).standard_normal((65536, 32))
t = time.perf_counter(); kmc.Sampler(kmc.GaussianIso(), 65536, 32, 10).close(); print(f"menu density, first sampler (process warm-up included): {1e3*(time.perf_counter()-t):8.1f} ms")
for label, make in [("ExprDensity", lambda: kmc.ExprDensity("-0.5*x*x")), ("CDensity", lambda: kmc.CDensity("double s = 0; for (int i = 0; i < n; ++i) s += x[i]*x[i]; return -0.5*s;"))]:
    t = time.perf_counter(); pdf = make(); t1 = time.perf_counter()
    s = kmc.Sampler(pdf, 65536, 32, 10); t2 = time.perf_counter(); s.close()
    s = kmc.Sampler(pdf, 65536, 32, 10); t3 = time.perf_counter(); s.close()
    print(f"{label:12s} create {1e3*(t1-t):8.1f} ms | first sampler {1e3*(t2-t1):8.1f} ms | second sampler {1e3*(t3-t2):8.1f} ms")
