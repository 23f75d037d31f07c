"""The phases of the drop-in call on a mid-size ensemble, timed one by one (what kmc.emcee does, in its order).
Usage (GPU box): python scripts/exp/dropin_phases_mid.py [walkers] [ndim] [generations]"""
import sys, time, threading
import numpy as np
sys.path.insert(0, '.')
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd import _lib
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
th = np.random.default_rng(0).standard_normal((nw, nd))
kmc.emcee(kmc.GaussianIso(), th[:256], niter=256 * 200, use_progress_meter=False, seed=1)
L = _lib.lib()
for rep in range(3):
    T = {}
    t0 = time.perf_counter()
    s = kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, 1, 2.0, 5, store_chain=True, store_logp=True, chain_by_walker=True); T["create"] = time.perf_counter() - t0
    t = time.perf_counter(); s.set_positions(th); T["set_positions"] = time.perf_counter() - t
    t = time.perf_counter(); out = (np.empty((nw, G // 2, nd)), np.empty((nw, G // 2)))
    warm = threading.Thread(target=lambda: [L.kmc_host_prefault(a.ctypes.data, a.nbytes, 4) for a in out], daemon=True); warm.start(); T["allocate + start prefault"] = time.perf_counter() - t
    t = time.perf_counter(); s.run(G); T["run (enqueue)"] = time.perf_counter() - t
    t = time.perf_counter(); s.sync(); T["sync"] = time.perf_counter() - t
    t = time.perf_counter(); warm.join(); T["join prefault"] = time.perf_counter() - t
    t = time.perf_counter(); s.chain(by_walker=True, out=out); T["chain"] = time.perf_counter() - t
    t = time.perf_counter(); s.accept_ratio(); T["accept_ratio"] = time.perf_counter() - t
    t = time.perf_counter(); s.close(); T["close"] = time.perf_counter() - t
    T["total"] = time.perf_counter() - t0
    print(f"{nw} x {nd}, {G} generations: " + " | ".join(f"{k} {v * 1e3:.2f}" for k, v in T.items()) + " ms", flush=True)
t = time.perf_counter(); kmc.emcee(kmc.GaussianIso(), th, niter=nw * G, use_progress_meter=False, seed=5); print(f"kmc.emcee: {(time.perf_counter() - t) * 1e3:.2f} ms")
