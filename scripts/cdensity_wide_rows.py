import sys, time
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
body = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"
for nw, nd, G in ((8192, 128, 128), (8192, 100, 128), (4096, 256, 64), (8192, 65, 128), (65536, 32, 256)):
    th = np.random.default_rng(1).standard_normal((nw, nd))
    for plan in ("", "generic"):
        import os
        if plan: os.environ["KMC_PLAN"] = plan
        else: os.environ.pop("KMC_PLAN", None)
        t0 = time.time()
        with kmc.Sampler(kmc.CDensity(body + f" // {plan}"), nw, nd, 2 * G, 0, 1, 2.0, 3, moments=True) as s:
            tc = time.time() - t0
            s.set_positions(th)
            s.run(G); s.sync()
            s.run(G); s.sync()
            print(f"{nw} x {nd} [{plan or 'staged'}]: {1e3 * s.last_run_ms() / (2 * G):8.3f} us per half-step (create {tc:.1f} s) {s.describe()[:60]}", flush=True)
