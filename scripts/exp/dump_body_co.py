"""Leave the code objects built ON THE GPU BOX for the C2 general-body vector kernel (second-neighbour coupling, evaluated per walker from
the LDS tile) under gpurun_out/co_cache_body (to be disassembled in the build container)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["KMC_CACHE_DIR"] = os.path.join(ROOT, "gpurun_out", "co_cache_body")
import kissmcmc_jl_amd as kmc

BODY = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; for (int i = 0; i + 2 < n; ++i) s += p[0] * x[i] * x[i + 2]; return -0.5 * s;"
with kmc.Sampler(kmc.CDensity(BODY, params=[0.01]), 65536, 32, 10 ** 9, 0, 1, 2.0, 7, moments=True) as s:
    s.set_positions(np.random.default_rng(0).standard_normal((65536, 32)))
    s.run(128)
    s.sync()
    print(s.describe())
