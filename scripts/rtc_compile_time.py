import time, sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
body = "double s = 0; for (int i = 0; i < n; ++i) s += x[i]*x[i]; return -0.5*s + %s;"
for k, (nw, nd) in enumerate([(100, 2), (100, 8), (1000, 4), (100, 32), (4096, 8)]):
    t0 = time.time()
    pdf = kmc.CDensity(body % f"{k}e-9")
    with kmc.Sampler(pdf, nw, nd, 64, 0, 1, 2.0, 1) as s:
        t1 = time.time()
        s.set_positions(np.random.default_rng(0).standard_normal((nw, nd)))
        s.run(64); s.sync()
        print(f"{nw} x {nd}: create {t1 - t0:.2f} s -- {s.describe()[:90]}")
