"""Very large ensembles against the oracle (index arithmetic beyond 2^31 bytes / 2^22 rows): a few generations,
bit-identical positions and counters.  Usage (GPU box): python scripts/big_parity.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc
import oracle

for nw, nd, G in ((1 << 22, 64, 6), (1 << 24, 2, 6), (1 << 18, 1024, 3), (3 * (1 << 20) + 2, 33, 5),
                  (1 << 22, 136, 3)):      # the last: 4.25 GiB of rows, byte offsets beyond 2^32
    rng = np.random.default_rng(nd)
    th = rng.standard_normal((nw, nd))
    t0 = time.time()
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, 1, 1, 2.0, 99, nthreads=os.cpu_count()), th,
                       store_chain=False)
    t1 = time.time()
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 1, 1, 2.0, 99, moments=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        ms = s.last_run_ms()
        pos, nacc = s.positions(), s.naccept()
        msum, msq, n = s.moments()
        how = s.describe()
    ok = np.array_equal(pos, ref["final_pos"]) and np.array_equal(nacc, ref["naccept"]) and n == ref["nmoment"] \
        and np.allclose(msum, ref["sum"], rtol=1e-10, atol=1e-7)
    print(f"{nw} x {nd} ({nw * nd * 8 / 2**30:.2f} GiB), {G} generations: {'bit-identical to the oracle' if ok else 'MISMATCH'}; "
          f"GPU {ms:.1f} ms, oracle {t1 - t0:.1f} s; {how[:60]}", flush=True)
    if not ok:
        sys.exit(1)
