"""Ragged rows against exact-size rows of the vector kernels: us per half-step by ndim around the geometry boundaries (ndim = 2 L K - 1 has 64-byte-aligned rows, ld = 2 L K, and the masked code;
ndim = 2 L K - 2 has unaligned rows too).   python scripts/probes/ragged_launch.py"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
for nw, nd in ((65536, 28), (65536, 30), (65536, 31), (65536, 32), (16384, 56), (16384, 60), (16384, 62), (16384, 63), (16384, 64), (8192, 112), (8192, 120), (8192, 126), (8192, 127), (8192, 128), (32768, 24), (32768, 31), (32768, 32)):
    for launch in (None,):
        if launch: os.environ["KMC_LAUNCH"] = launch
        else: os.environ.pop("KMC_LAUNCH", None)
        G = 4096
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, 1, 2.0, 5, moments=True) as s:
            th = np.random.default_rng(1).standard_normal((nw, nd))
            s.set_positions(th); s.run(G); s.sync()
            s.set_positions(th); s.run(G); s.sync()
            print(f"{nw} x {nd} KMC_LAUNCH={launch}: {s.last_run_ms() * 1e3 / (2 * G):.3f} us   {s.describe().split('(exact): ')[1][:200]}", flush=True)
