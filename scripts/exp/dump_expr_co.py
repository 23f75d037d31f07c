"""Leave the code objects hiprtc builds ON THE GPU BOX for the C2 ExprDensity vector kernel under gpurun_out/co_cache (to be
disassembled in the build container)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["KMC_CACHE_DIR"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "co_cache")
import kissmcmc_jl_amd as kmc

with kmc.Sampler(kmc.ExprDensity("-0.5*((x-p[0])*p[1])*((x-p[0])*p[1])", None, [0.0, 1.0]), 65536, 32, 10 ** 9, 0, 1, 2.0, 7, moments=True) as s:
    s.set_positions(np.random.default_rng(0).standard_normal((65536, 32)))
    s.run(128)
    s.sync()
    print(s.describe())
