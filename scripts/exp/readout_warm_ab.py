"""Chain read-out by walker into (a) a fresh numpy array, (b) a fresh array faulted in first (kmc_host_prefault), (c) the same array again
(warm); and the parts of kmc.emcee's wall time.  Usage (GPU box): python scripts/exp/readout_warm_ab.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd import _lib
L = _lib.lib()
for nw, nd, G in ((4096, 4, 2000), (4096, 32, 2000)):
    th = np.random.default_rng(0).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, 1, 2.0, 5, store_chain=True, store_logp=True) as s:
        s.set_positions(th); s.run(G); s.sync()
        K = G // 2
        def fresh():
            return np.empty((nw, K, nd)), np.empty((nw, K))
        s.chain(by_walker=True)
        o = fresh(); t = time.perf_counter(); s.chain(by_walker=True, out=o); t_cold = time.perf_counter() - t
        o = fresh(); t = time.perf_counter(); [L.kmc_host_prefault(a.ctypes.data, a.nbytes, 4) for a in o]; t_pf = time.perf_counter() - t
        t = time.perf_counter(); s.chain(by_walker=True, out=o); t_pre = time.perf_counter() - t
        t = time.perf_counter(); s.chain(by_walker=True, out=o); t_warm = time.perf_counter() - t
        t = time.perf_counter(); s.chain(by_walker=True, out=o); t_warm2 = time.perf_counter() - t
        print(f"{nw} x {nd}, {K} samples/walker ({(o[0].nbytes + o[1].nbytes) / 1e6:.0f} MB): fresh arrays {t_cold*1e3:6.2f} ms | prefault (4 threads) {t_pf*1e3:6.2f} ms, then read-out {t_pre*1e3:6.2f} ms | "
              f"same arrays again {t_warm*1e3:6.2f}, {t_warm2*1e3:6.2f} ms", flush=True)
