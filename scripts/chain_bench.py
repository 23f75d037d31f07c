"""C2 with the thinned chain kept on the device ("one run with nthin such that the chain fits"):
65 536 walkers x 32 dims, 10^4 generations, burn-in 5 000, nthin = 100 -> 50 stored samples per walker
(839 MB of chain + 26 MB of log-densities).  Prints the device loop time, the per-half-step period and the D2H time.
Usage (GPU box): python scripts/chain_bench.py [nthin]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc


def main():
    nthin = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    nw, nd, G, nburn = 65536, 32, 10000, 5000
    th = np.random.default_rng(0).standard_normal((nw, nd))
    for store in (False, True):
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, nthin, 2.0, 11, store_chain=store, store_logp=store, moments=True) as s:
            s.set_positions(th)
            s.run(G)
            s.sync()
            s.set_positions(th)
            s.run(G)
            s.sync()
            ms = s.last_run_ms()
            line = f"C2 nthin={nthin} chain={'on ' if store else 'off'}: {ms:8.2f} ms device loop, {ms / (2 * G) * 1e3:.3f} us per half-step, {nw * G / ms / 1e6:.3f} G walker-steps/s"
            if store:
                t0 = time.perf_counter()
                ch, cl = s.chain()
                dt = time.perf_counter() - t0
                line += f"; chain {ch.shape} = {ch.nbytes / 1e6:.0f} MB + logp {cl.nbytes / 1e6:.0f} MB to the host in {dt * 1e3:.0f} ms ({(ch.nbytes + cl.nbytes) / dt / 1e9:.1f} GB/s)"
                assert ch.shape == (s.nsamples, nw, nd) and np.isfinite(ch).all()
            print(line, flush=True)


if __name__ == "__main__":
    main()
