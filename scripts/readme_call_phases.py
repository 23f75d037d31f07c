"""Where the wall time of the README call goes (100 walkers x 1-D, niter = 10^5): phases of api.emcee timed one by one."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc

pdf = kmc.Exponential(1.0)
theta0s = np.asarray(kmc.make_theta0s(0.5, 0.1, pdf, 100, rng=3), dtype=np.float64).reshape(100, 1)
kmc.emcee(pdf, theta0s[:, 0], niter=10 ** 5, seed=7, use_progress_meter=False)          # warm: library, HIP context
for rep in range(3):
    t = [time.perf_counter()]
    s = kmc.Sampler(pdf, 100, 1, 1000, 500, 1, 2.0, 7, store_chain=True, store_logp=True, chain_by_walker=True)
    t.append(time.perf_counter())
    s.set_positions(theta0s)
    t.append(time.perf_counter())
    s.run(1000)
    t.append(time.perf_counter())
    s.sync()
    t.append(time.perf_counter())
    ch, lp = s.chain(by_walker=True)
    t.append(time.perf_counter())
    ar = s.accept_ratio()
    t.append(time.perf_counter())
    s.close()
    t.append(time.perf_counter())
    names = ["create", "set_positions", "run (enqueue)", "sync", "chain download", "accept_ratio", "destroy"]
    print("  ".join(f"{n} {1e3 * (b - a):.3f}" for n, a, b in zip(names, t[:-1], t[1:])), f" total {1e3 * (t[-1] - t[0]):.3f} ms", flush=True)
t0 = time.perf_counter(); kmc.emcee(pdf, theta0s[:, 0], niter=10 ** 5, seed=7, use_progress_meter=False); print(f"api.emcee: {1e3 * (time.perf_counter() - t0):.3f} ms")
