"""Run one BASELINE config twice for a number of generations -- warm-up, then timed (a plain target for rocprofv3).
Usage: python3 scripts/run_cfg.py C5 [generations] [moments 0/1]
HBM32 / HBM128: the HBM-resident shapes of bench.py's other_configs (state 512 MiB; initial ensemble drawn on the device);
MID4K / MID16K: 4096 / 16384 walkers x 4 dims (bench.py's MID_* lines); MID8Kx64 / MID16Kx32: 4 MiB of state."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc

CONFIGS = {
    "C2": (kmc.GaussianIso(), 65536, 32),
    "C3": (kmc.Rosenbrock(), 16384, 64),
    "C5": (kmc.GaussianIso(), 8192, 1024),
    "C1": (kmc.Exponential(), 100, 1),
    "HBM32": (kmc.GaussianIso(), 2097152, 32),
    "HBM128": (kmc.GaussianIso(), 524288, 128),
    "MID4K": (kmc.GaussianIso(), 4096, 4),          # mid-size ensembles with short rows: one launch per generation (KMC_DEBUG=fused=0: two)
    "MID16K": (kmc.GaussianIso(), 16384, 4),
    "MID8Kx64": (kmc.GaussianIso(), 8192, 64),      # 4 MiB of state: the lane-striped one-launch-per-generation form since round 5 (KMC_DEBUG=fused=0: two launches)
    "MID16Kx32": (kmc.GaussianIso(), 16384, 32),
}
name = sys.argv[1]
G = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mom = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
pdf, nw, nd = CONFIGS[name]
with kmc.Sampler(pdf, nw, nd, 10 ** 9, 0, 1, 2.0, 7, moments=mom) as s:
    if name.startswith("HBM"):
        s.init_ball(np.zeros(nd), np.ones(nd), seed=7)
    else:
        rng = np.random.default_rng(0)
        th = rng.standard_normal((nw, nd))
        if name == "C1":
            th = 0.5 + 0.1 * np.abs(th)
        if name == "C3":
            th *= 0.1
        s.set_positions(th)
    s.run(G)            # warm-up: code objects, graph instantiation, launch-mode measurement
    s.sync()
    s.run(G)            # timed (HIP events on the sampler's stream); a trace's second half of dispatches is this run
    s.sync()
    print(name, "generations", G, "moments", int(mom), "ms", s.last_run_ms(), "us/half-step", s.last_run_ms() / (2 * G) * 1e3)
    print(s.describe())
