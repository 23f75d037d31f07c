import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import kissmcmc_jl_amd as kmc
import torch
for nw, nd, G in ((4096, 4, 2000), (16384, 4, 2000), (4096, 32, 2000)):
    th = np.random.default_rng(0).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, 1, 2.0, 5, store_chain=True, store_logp=True) as s:
        s.set_positions(th); s.run(G); s.sync()
        for rep in range(2):
            t = time.perf_counter(); a, la = s.chain(by_walker=True); t_bw = time.perf_counter() - t
            t = time.perf_counter(); b, lb = s.chain(); t_sm = time.perf_counter() - t
        # raw: pinned destination, plain hipMemcpy of the same bytes
        n = a.size
        pin = torch.empty(n, dtype=torch.float64).pin_memory()
        dev = torch.empty(n, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize(); t = time.perf_counter(); pin.copy_(dev); torch.cuda.synchronize(); t_pin = time.perf_counter() - t
        page = np.empty(n)
        t = time.perf_counter(); np.copyto(page, a.reshape(-1)); t_np = time.perf_counter() - t
        print(f"{nw} x {nd}, {G//2} samples/walker, chain {a.nbytes/1e6:.0f} MB + logp {la.nbytes/1e6:.0f} MB: by walker {t_bw*1e3:6.2f} ms | sample-major {t_sm*1e3:6.2f} ms | "
              f"D2H of the chain bytes into pinned memory {t_pin*1e3:6.2f} ms | numpy copy of that many bytes (1 thread) {t_np*1e3:6.2f} ms", flush=True)
