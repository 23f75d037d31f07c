"""Wall time of the drop-in call kmc.emcee(...) at C2 with a thinned chain, by part: sampling, chain read-out in the
reference's order (device transposition, kmc_sampler_get_chain_by_walker) against sample-major read-out + numpy reorder.
Usage (GPU box): python scripts/exp/dropin_walltime.py [nthin]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc

nthin = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nw, nd, G = 65536, 32, 10000
th = np.random.default_rng(0).standard_normal((nw, nd))
kmc.emcee(kmc.GaussianIso(), th[:256], niter=256 * 200, use_progress_meter=False, seed=1)      # warm the process
t = time.perf_counter()
thetas, acc, logd, _ = kmc.emcee(kmc.GaussianIso(), th, niter=nw * G, nthin=nthin, use_progress_meter=False, seed=5)
t_all = time.perf_counter() - t
print(f"kmc.emcee 65536 x 32, {G} generations, nthin {nthin}: {t_all * 1e3:8.1f} ms wall, chain {thetas.nbytes / 1e6:.0f} MB, thetas{thetas.shape} accept {acc.mean():.3f}", flush=True)
with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, nthin, 2.0, 5, store_chain=True, store_logp=True) as s:
    t = time.perf_counter(); s.set_positions(th); t_set = time.perf_counter() - t
    t = time.perf_counter(); s.run(G); s.sync(); t_run = time.perf_counter() - t
    t = time.perf_counter(); a, la = s.chain(by_walker=True); t_bw = time.perf_counter() - t
    t = time.perf_counter(); b, lb = s.chain(); t_sm = time.perf_counter() - t
    t = time.perf_counter(); c = np.ascontiguousarray(b.transpose(1, 0, 2)); lc = np.ascontiguousarray(lb.T); t_np = time.perf_counter() - t
    assert np.array_equal(a, c) and np.array_equal(la, lc) and np.array_equal(a, thetas)
    print(f"  set_positions {t_set * 1e3:7.1f} ms | run {t_run * 1e3:7.1f} ms | chain by walker (device transposition + D2H) {t_bw * 1e3:7.1f} ms "
          f"| sample-major D2H {t_sm * 1e3:7.1f} ms + numpy reorder {t_np * 1e3:7.1f} ms")
t = time.perf_counter(); sq = kmc.squash_walkers(thetas, acc, logd, verbose=False); t_sq = time.perf_counter() - t
print(f"  squash_walkers (walker-major: a view) {t_sq * 1e3:7.2f} ms -> {sq[0].shape}")
t = time.perf_counter()
thetas2, acc2, logd2, _ = kmc.emcee(kmc.GaussianIso(), th, niter=nw * G, nthin=nthin, use_progress_meter=False, seed=5, stream_chain=True)
t_st = time.perf_counter() - t
assert np.array_equal(thetas2, thetas) and np.array_equal(logd2, logd)
print(f"kmc.emcee(..., stream_chain=True): chain streamed by walker into page-locked host arrays while sampling: {t_st * 1e3:8.1f} ms wall")
