import os, sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
SHAPES = ((4096, 1), (4096, 4), (16384, 4), (32768, 4), (65536, 2), (4096, 8), (4096, 12), (4096, 32), (8192, 32), (2050, 128))
if len(sys.argv) > 1:           # e.g. 4096x6,8192x8
    SHAPES = tuple(tuple(int(v) for v in item.split("x")) for item in sys.argv[1].split(","))
for nw, nd in SHAPES:
    G = 8192 if nw <= 16384 else 2048
    best = 1e9
    for rep in range(3):
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, 2 * G, 0, 1, 2.0, 3, moments=True) as s:
            s.set_positions(np.random.default_rng(1).standard_normal((nw, nd)))
            s.run(G); s.sync(); s.run(G); s.sync()
            best = min(best, 1e3 * s.last_run_ms() / (2 * G)); how = s.describe()
    print(f"{nw:7d} x {nd:3d}  {best:6.3f} us per half-step  {'one' if 'one launch' in how else 'two'}", flush=True)
