"""Ragged rows against exact-size rows of the vector kernels: us per half-step by ndim around the geometry boundaries (ndim = 2 L K - 1 has 64-byte-aligned rows, ld = 2 L K;
ndim = 2 L K - 2 has unaligned rows too).   python scripts/probes/ragged_launch.py [reps]      (KMC_LIB_PATH=<variant build> for a same-box A/B: scripts/ab.sh)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
out = []
for nw, nd in ((65536, 28), (65536, 30), (65536, 31), (65536, 32), (16384, 56), (16384, 60), (16384, 63), (16384, 64), (8192, 112), (8192, 120), (8192, 127), (8192, 128), (32768, 24), (32768, 31), (32768, 32),
               (65536, 20), (16384, 40), (8192, 80), (4096, 200), (4096, 256),
               (4096, 100), (4096, 128), (8192, 60), (8192, 64), (16384, 31), (16384, 32), (4096, 20), (2048, 250), (2048, 256)):     # (the last rows: one launch per generation)
    G = 4096
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, 10 ** 9, 0, 1, 2.0, 5, moments=True) as s:
        th = np.random.default_rng(1).standard_normal((nw, nd))
        s.set_positions(th); s.run(G); s.sync()
        ts = []
        for r in range(REPS):
            s.run(G); s.sync(); ts.append(s.last_run_ms() * 1e3 / (2 * G))
        geo = s.describe().split('(exact): ')[-1].split(', hipGraph')[0][:90]
        print(f"{nw:6d} x {nd:4d}: {min(ts):.3f} us   {geo}", flush=True)
