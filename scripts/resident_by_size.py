import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
for nw in (32, 64, 66, 100, 128, 130, 256, 512, 1000, 1024, 1026, 1500, 2048, 2050, 4096):
    for nd in (1, 4):
        pdf = kmc.Exponential(1.0) if nd == 1 else kmc.GaussianIso(0.0, 1.0)
        G = 20000
        with kmc.Sampler(pdf, nw, nd, G, 0, 1, 2.0, 3, moments=True) as s:
            s.set_positions(0.5 + 0.1 * np.abs(np.random.default_rng(1).standard_normal((nw, nd))))
            s.run(G); s.sync()
            s.run(G); s.sync()
            print(f"{nw:5d} x {nd}: {1e3 * s.last_run_ms() / (2 * G):.3f} us per half-step", flush=True)
