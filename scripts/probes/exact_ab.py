"""Exact-size rows at the C2 shape, menu and runtime-compiled densities: us per half-step (G generations timed, the faster of three) -- run once per library build
(KMC_LIB_PATH=<variant build>) on ONE box to tell a code change from box-to-box noise.   python scripts/probes/exact_ab.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
body = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"
two_sums = "double s = 0, t = 0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += x[i]; } return -0.5 * (s + p[0] * t * t);"
coupled = ("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; double c = 0; for (int i = 0; i + 2 < n; ++i) c += x[i] * x[i + 2]; return -0.5 * s - p[0] * c;")
nw, nd, G = 65536, 32, 2000
th = np.random.default_rng(2).standard_normal((nw, nd))
out = []
for name, pdf, kw in (("menu", kmc.GaussianIso(), {}), ("menu, chain on", kmc.GaussianIso(), dict(store_chain=True, nthin=100)), ("body: sum", kmc.CDensity(body), {}),
                      ("body: two sums", kmc.CDensity(two_sums, params=[0.05]), {}), ("body: coupled", kmc.CDensity(coupled, params=[0.2]), {})):
    nthin = kw.pop("nthin", 1)
    with kmc.Sampler(pdf, nw, nd, 4 * G, G, nthin, 2.0, 3, moments=True, **kw) as s:
        s.set_positions(th)
        s.run(G); s.sync()
        ts = []
        for r in range(3):
            s.run(G); s.sync(); ts.append(s.last_run_ms() * 1e3 / (2 * G))
    out.append(f"{name}: {min(ts):.3f}")
print("; ".join(out), flush=True)
