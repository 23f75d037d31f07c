"""C2 with the thinned chain stored: 65 536 walkers x 32 dims, 10^4 generations, burn-in 5 000.
Three ways per nthin: chain off; chain kept on the device (downloaded afterwards); chain STREAMED to page-locked host
memory while sampling (KMC_STREAM_CHAIN: device ring of three blocks, second stream).  Prints the device loop time, the
per-half-step period, wall time to the point where the chain is in host memory, and the implied host-bound rate.
Usage (GPU box): python scripts/chain_bench.py [nthin ...]        (default: 100 40 10)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc


def main():
    nthins = [int(a) for a in sys.argv[1:]] or [100, 40, 10]
    nw, nd, G, nburn = 65536, 32, 10000, 5000
    th = np.random.default_rng(0).standard_normal((nw, nd))
    ch = cl = None
    for nthin in nthins:
        ns = (G - nburn) // nthin
        gb = ns * nw * (nd + 1) * 8 / 1e9
        for mode in ("off", "device", "stream"):
            store = mode != "off"
            with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, nthin, 2.0, 11, store_chain=store, store_logp=store, moments=True,
                             stream_chain=mode == "stream") as s:
                s.set_positions(th)
                s.run(G)
                s.sync()
                s.set_positions(th)
                t0 = time.perf_counter()
                s.run(G)
                s.sync()
                t_run = time.perf_counter() - t0
                ms = s.last_run_ms()
                line = (f"C2 nthin={nthin:4d} chain={mode:6s}: device loop {ms:8.2f} ms = {ms / (2 * G) * 1e3:.3f} us per half-step, "
                        f"{nw * G / ms / 1e6:.3f} G walker-steps/s (loop); run+sync wall {t_run * 1e3:8.1f} ms")
                if store:
                    t1 = time.perf_counter()
                    ch, cl = s.chain()
                    dt = time.perf_counter() - t1
                    assert ch.shape == (ns, nw, nd) and np.isfinite(ch[-1]).all() and np.isfinite(ch[0]).all()
                    total = t_run + dt
                    line += (f"; chain {gb:.2f} GB in host memory {total * 1e3:8.1f} ms after the start "
                             f"({nw * G / total / 1e9:.3f} G walker-steps/s end to end, {gb / total:.1f} GB/s of samples)")
                    if mode == "stream":
                        line += f"  [{s.describe().split(';')[-1].strip()}]"
                print(line, flush=True)
                ch = cl = None          # release the host arrays here, outside the next timing window (freeing 8 GB takes 0.3 s)


if __name__ == "__main__":
    main()
