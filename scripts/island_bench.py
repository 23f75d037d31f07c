"""Island-mode throughput at the C2 shape for island sizes / epoch lengths."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc

nw, nd = 65536, 32
th = np.random.default_rng(0).standard_normal((nw, nd))
for S in (64, 128, 256):
    for k in (8, 32, 128):
        for mom in (True, False):
            with kmc.Sampler(kmc.GaussianIso(), nw, nd, 10 ** 9, 0, 1, 2.0, 1, moments=mom, island_gens=k, island_size=S) as s:
                s.set_positions(th)
                s.run(256)
                s.sync()
                ts = []
                for _ in range(3):
                    s.run(1024)
                    s.sync()
                    ts.append(s.last_run_ms())
                t = min(ts)
                acc = s.naccept().mean() / s.generation
                print(f"S={S:3d} epoch={k:3d} moments={int(mom)}  {t / 1024 * 1e3:6.2f} us/generation  {nw * 1024 / t / 1e6:7.2f} Gsteps/s  acc={acc:.3f}", flush=True)
