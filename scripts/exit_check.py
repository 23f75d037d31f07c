import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
cd = kmc.CDensity("return -0.5 * x[0] * x[0];")
s = kmc.Sampler(cd, 100, 1, 100, store_chain=True)
s.set_positions(np.random.default_rng(0).standard_normal((100, 1)))
s.run(100)
th = kmc.emcee(cd, np.zeros(100) + 0.1 * np.arange(100), niter=10**4, seed=1, use_progress_meter=False)
print("ok, leaving with a live sampler and density")
