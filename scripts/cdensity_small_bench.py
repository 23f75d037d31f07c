"""A user-written C++ density (CDensity) on the reference's own problem sizes: resident mode (one workgroup, one walker per thread,
many generations per launch) against the multi-launch kernels (KMC_DEBUG=no-resident).  python scripts/cdensity_small_bench.py"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import kissmcmc_jl_amd as kmc
    rosen = ("double s = 0.0; for (int i = 0; i + 1 < n; ++i) { double d = x[i + 1] - x[i] * x[i]; double e = p[0] - x[i]; s += p[1] * (d * d) + e * e; } "
             "return -(s * (1.0 / p[2]));")
    for nw, nd, G in ((100, 2, 20000), (100, 8, 20000), (1000, 4, 5000)):
        pdf = kmc.CDensity(rosen, params=[1.0, 100.0, 20.0])
        th = 0.1 * np.random.default_rng(1).standard_normal((nw, nd))
        with kmc.Sampler(pdf, nw, nd, G, G // 2, 1, 2.0, 3, store_chain=True, store_logp=True) as s:
            s.set_positions(th)
            s.run(200)
            s.sync()
            s.set_positions(th)
            t0 = time.perf_counter()
            s.run(G)
            s.sync()
            wall = time.perf_counter() - t0
            ms = s.last_run_ms()
            print(f"{nw:5d} x {nd:2d} Rosenbrock as a CDensity, {G} generations: {ms * 1e3 / (2 * G):7.3f} us per half-step (device), {nw * G / wall:.3e} walker-steps/s wall, "
                  f"{s.launch_count} launches -- {s.describe().split(',')[0]}")
elif len(sys.argv) > 1 and sys.argv[1] == "menu":
    # the same small problems with the MENU densities next to their CDensity restatements, both resident (KMC_DEBUG=resident=pair: the
    # two-lanes-per-walker kernel instead of one walker per thread for ndim <= 8)
    import kissmcmc_jl_amd as kmc
    cases = [("exponential 100 x 1", kmc.Exponential(), kmc.CDensity("return x[0] < 0.0 ? -INFINITY : -x[0];"), 100, 1),
             ("gaussian 100 x 2", kmc.GaussianIso(), kmc.CDensity("return -0.5 * (x[0] * x[0] + x[1] * x[1]);"), 100, 2),
             ("gaussian 100 x 8", kmc.GaussianIso(), kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"), 100, 8),
             ("gaussian 100 x 32", kmc.GaussianIso(), kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"), 100, 32),
             ("gaussian 1000 x 4", kmc.GaussianIso(), kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"), 1000, 4)]
    for label, menu, body, nw, nd in cases:
        th = 0.5 + 0.1 * np.abs(np.random.default_rng(1).standard_normal((nw, nd)))
        out = []
        for pdf in (menu, body):
            with kmc.Sampler(pdf, nw, nd, 8192, 0, 1, 2.0, 3) as s:
                s.set_positions(th)
                s.run(4096)
                s.sync()
                s.run(4096)
                s.sync()
                out.append(s.last_run_ms() * 1e3 / 8192)
        print(f"{label:22s}: menu density {out[0]:6.3f} us per half-step, CDensity {out[1]:6.3f}   [KMC_DEBUG={os.environ.get('KMC_DEBUG', '')}]")
else:
    for env in ({}, {"KMC_DEBUG": "no-resident"}):
        print("==", env or "default")
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, **env))
