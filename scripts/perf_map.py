"""Throughput map (walker-steps/s, device time) over ensemble shapes: the menu Gaussian and the same density written as a function body."""
import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc

body = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"
NW = (100, 1000, 2048, 4096, 16384, 65536, 262144)
ND = (1, 4, 32, 128) if '--ragged' not in sys.argv else (3, 10, 20, 31, 50, 100, 200)      # --ragged: row lengths that are not powers of two (what callers mostly have)
general = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; for (int i = 0; i + 2 < n; ++i) s += p[0] * x[i] * x[i + 2]; return -0.5 * s;"
for name, make in (("menu GaussianIso", lambda: kmc.GaussianIso()), ("CDensity, a sum over elements (recognised: lane-striped)", lambda: kmc.CDensity(body)),
                   ("CDensity, a general body (second-neighbour coupling, two loops: rows lane-striped, evaluated per walker)", lambda: kmc.CDensity(general, params=[0.2]))):
    print(f"\n{name}: walker-steps/s (us per half-step) [mode]")
    print("walkers \\ ndim | " + " | ".join(f"{d:>24d}" for d in ND))
    for nw in NW:
        cells = []
        for nd in ND:
            if nw < nd + 2:
                cells.append(f"{'-':>24s}"); continue
            G = int(max(64, min(20000, 4e7 / (nw * max(nd, 8) / 8))))
            G -= G % 64
            with kmc.Sampler(make(), nw, nd, 2 * G, 0, 1, 2.0, 3, moments=True) as s:
                s.set_positions(np.random.default_rng(1).standard_normal((nw, nd)))
                s.run(G); s.sync()
                s.run(G); s.sync()
                ms = s.last_run_ms()
                how = s.describe()
            mode = ("res2" if "two walkers" in how else "res" if "resident" in how else "one" if "generation_lane" in how else "oneg" if "generation_group" in how else
                    "vec" if "half_step_vec" in how else "stg" if "staged" in how else "gen")
            cells.append(f"{nw * G / (ms * 1e-3):9.2e} ({1e3 * ms / (2 * G):6.2f}) {mode:>4s}")
        print(f"{nw:>14d} | " + " | ".join(cells), flush=True)
