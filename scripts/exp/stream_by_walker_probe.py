"""Where the time of a streamed chain goes: sampler creation (page-locking the host arrays), the loop, the final sync;
sample-major (DMA copies) against by-walker (copy kernel writing into the host arrays)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc
nthin = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nw, nd, G = 65536, 32, 10000
th = np.random.default_rng(0).standard_normal((nw, nd))
for label, kw in [("device chain", dict()), ("streamed sample-major", dict(stream_chain=True)), ("streamed by walker", dict(stream_chain=True, chain_by_walker=True))] * 2:
    t0 = time.perf_counter()
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, nthin, 2.0, 5, store_chain=True, store_logp=True, **kw) as s:
        t1 = time.perf_counter(); s.set_positions(th)
        t2 = time.perf_counter(); s.run(G); t3 = time.perf_counter(); s.sync(); t4 = time.perf_counter()
        loop_ms = s.last_run_ms()
        a, la = s.chain(by_walker=True); t5 = time.perf_counter()
    print(f"{label:24s} create {1e3*(t1-t0):7.1f} | set_positions {1e3*(t2-t1):6.1f} | enqueue {1e3*(t3-t2):6.1f} | sync {1e3*(t4-t3):7.1f} (device loop {loop_ms:7.1f}) | chain(by_walker) {1e3*(t5-t4):7.1f} | total {1e3*(t5-t0):7.1f} ms", flush=True)
