"""A general function body (coupled Gaussian) by row length: the vector kernel with per-walker evaluation against the one-walker-per-lane
kernels (KMC_DEBUG=no-body-vec).  python scripts/exp/body_vec_rows.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc

# a body no per-element form can express (second-neighbour coupling: two loops) -- and, for comparison, a two-sum body the library recognises
BODY = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; for (int i = 0; i + 2 < n; ++i) s += p[0] * x[i] * x[i + 2]; return -0.5 * s;"
TWO_SUMS = "double s = 0, t = 0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += x[i]; } return -0.5 * (s + p[0] * t * t);"
for nw, nd, G in ((65536, 32, 1024), (65536, 8, 1024), (16384, 64, 1024), (65536, 128, 512), (16384, 256, 256), (8192, 1024, 128), (4096, 32, 1024)):
    out = []
    for dbg in ("", "no-body-vec", "two-sums"):
        os.environ["KMC_DEBUG"] = "" if dbg == "two-sums" else dbg
        pdf = kmc.CDensity(TWO_SUMS if dbg == "two-sums" else BODY, params=[0.01])
        th = np.random.default_rng(0).standard_normal((nw, nd))
        with kmc.Sampler(pdf, nw, nd, 10 ** 9, 0, 1, 2.0, 7, moments=True) as s:
            s.set_positions(th)
            s.run(max(G, 832))
            s.sync()
            s.run(G)
            s.sync()
            out.append((s.last_run_ms() / (2 * G) * 1e3, s.describe().split(",")[0].split(": ")[1][:40]))
    print(f"{nw:6d} x {nd:4d}: general body in the vector kernel {out[0][0]:8.2f} us ({out[0][1]}), one walker per lane {out[1][0]:8.2f} us ({out[1][1]})  -> x{out[1][0] / out[0][0]:.2f};"
          f"  two-sum body (recognised, lane-striped) {out[2][0]:8.2f} us", flush=True)
