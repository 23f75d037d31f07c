"""One against two walkers per lane group at the planner's threshold (about 2048 single-walker waves per half), bench-shaped jobs, best of three: is the rule of rounds 1-2 still right now that the
draw ring is off for exact-size rows at ITER = 2?  (It is: C3 3.231 / 3.246 us, 32 768 x 32 3.07 / 3.33, 24 576 x 64 3.80 / 3.66 -- profiles/NOTES.md round 5.)   python scripts/probes/iter_ab.py"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
CASES = [("C3 rosen 16384x64", kmc.Rosenbrock, 16384, 64, 0.1, ("16,2,1", "16,2,2")), ("gauss 16384x64", kmc.GaussianIso, 16384, 64, 1.0, ("16,2,1", "16,2,2")),
         ("gauss 32768x32", kmc.GaussianIso, 32768, 32, 1.0, ("8,2,1", "8,2,2")), ("gauss 8192x128", kmc.GaussianIso, 8192, 128, 1.0, ("32,2,1", "32,2,2")),
         ("gauss 24576x64", kmc.GaussianIso, 24576, 64, 1.0, ("16,2,1", "16,2,2")), ("gauss 40960x32", kmc.GaussianIso, 40960, 32, 1.0, ("8,2,1", "8,2,2"))]
G = 4096
for name, pdf, nw, nd, scale, plans in CASES:
    out = []
    for plan in plans:
        os.environ["KMC_PLAN"] = plan
        best = 1e9
        for rep in range(3):
            with kmc.Sampler(pdf(), nw, nd, G, G // 2, 1, 2.0, 12345, moments=True) as s:
                th = scale * np.random.default_rng(12345).standard_normal((nw, nd))
                s.set_positions(th); s.run(G); s.sync()
                s.set_positions(th); s.run(G); s.sync()
                best = min(best, s.last_run_ms() * 1e3 / (2 * G))
                how = s.describe()
        out.append((plan, best, "ring" if "ring" in how else ""))
    print(name, " | ".join(f"{p}: {t:.3f} us" for p, t, _ in out), f"({out[1][1] / out[0][1]:.3f})", flush=True)
