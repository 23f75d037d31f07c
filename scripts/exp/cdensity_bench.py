"""Body densities (CDensity) against the menu and term / pair forms of the same log-density: us per half-step.
Usage (GPU box): python scripts/exp/cdensity_bench.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc

G = "double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"
R = "double s = 0.0; for (int i = 0; i + 1 < n; ++i) { double d = x[i + 1] - x[i] * x[i]; double e = 1.0 - x[i]; s += 100.0 * (d * d) + e * e; } return -s / 20.0;"
for name, nw, nd, forms in [("C2 65536x32 Gaussian", 65536, 32, [("menu", kmc.GaussianIso()), ("term/pair", kmc.ExprDensity("-0.5*x*x")), ("body", kmc.CDensity(G))]),
                            ("C3 16384x64 Rosenbrock", 16384, 64, [("menu", kmc.Rosenbrock()), ("term/pair", kmc.ExprDensity("d < n-1 ? -((1-x)*(1-x))/20 : 0.0", "-(100*((y-x*x)*(y-x*x)))/20")), ("body", kmc.CDensity(R))]),
                            ("4096x8 Gaussian", 4096, 8, [("menu", kmc.GaussianIso()), ("body", kmc.CDensity(G))])]:
    th = 0.3 * np.random.default_rng(0).standard_normal((nw, nd))
    for label, pdf in forms:
        with kmc.Sampler(pdf, nw, nd, 10 ** 9, 0, 1, 2.0, 7) as s:
            s.set_positions(th)
            s.run(512); s.sync()
            s.run(1024); s.sync()
            ms = s.last_run_ms()
            print(f"{name:24s} {label:10s} {ms / 2048 * 1e3:7.2f} us per half-step  {nw * 1024 / ms / 1e6:7.3f} G walker-steps/s   [{s.describe()[:60]}]", flush=True)
