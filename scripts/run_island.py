"""Island-mode run at the C2 shape (a plain target for rocprofv3).  Usage: python3 scripts/run_island.py [S] [epoch] [gens]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
nw, nd = 65536, 32
th = np.random.default_rng(0).standard_normal((nw, nd))
with kmc.Sampler(kmc.GaussianIso(), nw, nd, 10 ** 9, 0, 1, 2.0, 1, moments=True, island_gens=k, island_size=S) as s:
    s.set_positions(th)
    s.run(G)
    s.sync()
    print("island S", S, "epoch", k, "gens", G, "ms", s.last_run_ms(), "Gsteps/s", nw * G / s.last_run_ms() / 1e6)
