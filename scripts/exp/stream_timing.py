import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc
nw, nd, G, nburn, nthin = 65536, 32, 10000, 5000, int(sys.argv[1]) if len(sys.argv) > 1 else 10
th = np.random.default_rng(0).standard_normal((nw, nd))
t0 = time.perf_counter()
s = kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, nthin, 2.0, 11, store_chain=True, store_logp=True, moments=True, stream_chain=True)
print("create (incl. host arrays + page-locking)", time.perf_counter() - t0)
s.set_positions(th)
for rep in range(2):
    s.set_positions(th)
    t0 = time.perf_counter(); s.run(G); t1 = time.perf_counter(); s.sync(); t2 = time.perf_counter(); c, l = s.chain(); t3 = time.perf_counter()
    print(f"rep {rep}: enqueue {t1-t0:.3f} s, sync {t2-t1:.3f} s, chain() {t3-t2:.3f} s, device loop {s.last_run_ms():.1f} ms; {c.nbytes/1e9:.2f} GB -> {c.nbytes/1e9/(t2-t0):.1f} GB/s")
s.close()
