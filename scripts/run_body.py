"""A caller-written function body at the C2 shape, twice for a number of generations -- warm-up, then timed (a plain target for rocprofv3).
Usage: python3 scripts/run_body.py [coupled|sum|two-sums] [generations]
coupled: second-neighbour coupling in two loops (rows lane-striped, the body evaluated per walker from the wave's LDS tile);
sum / two-sums: bodies recognised as sums over elements (lane-striped like a menu density, no LDS)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc

BODIES = {
    "coupled": ("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; for (int i = 0; i + 2 < n; ++i) s += p[0] * x[i] * x[i + 2]; return -0.5 * s;", [0.01]),
    "sum": ("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;", []),
    "two-sums": ("double s = 0, t = 0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += x[i]; } return -0.5 * (s + p[0] * t * t);", [0.01]),
}
kind = sys.argv[1] if len(sys.argv) > 1 else "coupled"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 256
body, params = BODIES[kind]
nw, nd = 65536, 32
with kmc.Sampler(kmc.CDensity(body, params=params), nw, nd, 10 ** 9, 0, 1, 2.0, 7, moments=True) as s:
    s.set_positions(np.random.default_rng(0).standard_normal((nw, nd)))
    s.run(max(G, 832))          # warm-up: code objects, graph instantiation, launch-mode measurement
    s.sync()
    s.run(G)
    s.sync()
    print(kind, "generations", G, "ms", s.last_run_ms(), "us/half-step", s.last_run_ms() / (2 * G) * 1e3)
    print(s.describe())
