"""The draw ring of the vector kernels (HalfStepArgs::ring) forced off / on (KMC_DEBUG=ring=0|1) over ensemble shapes: us per half-step, menu Gaussian, moments on.
    python scripts/probes/ring_ab.py [--small]     (default: the large-ensemble geometries, ITER >= 4; --small: ITER 1-2, exact-size and ragged rows)   -> profiles/r05_ring_ab.txt"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
SMALL = "--small" in sys.argv
for nw, nd in (((16384, 64), (32768, 64), (65536, 64), (8192, 128), (16384, 128), (32768, 128), (16384, 60), (32768, 60), (8192, 120), (16384, 100), (32768, 50), (8192, 66), (16384, 36), (16384, 63), (8192, 127), (4096, 100), (32768, 127), (65536, 63)) if SMALL else ((65536, 128), (131072, 64), (262144, 64), (131072, 128), (524288, 64), (1048576, 64))):
    res = []
    for dbg in ("ring=0", "ring=1"):
        if dbg: os.environ["KMC_DEBUG"] = dbg
        else: os.environ.pop("KMC_DEBUG", None)
        G = 2048 if SMALL else 256
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, 3 * G, G, 1, 2.0, 5, moments=True) as s:
            s.init_ball(np.zeros(nd), np.ones(nd), seed=5)
            s.run(G); s.sync()
            s.run(G); s.sync(); a = s.last_run_ms() * 1e3 / (2 * G)
            s.run(G); s.sync(); b = s.last_run_ms() * 1e3 / (2 * G)
            how = s.describe().split(", hipGraph")[0].split("(exact): ")[1]
        res.append(min(a, b))
    print(f"{nw} x {nd} (state {nw * nd * 8 / 2**20:.0f} MiB) {how}: no ring {res[0]:.2f} us, ring {res[1]:.2f} us per half-step ({res[1] / res[0]:.3f})", flush=True)
