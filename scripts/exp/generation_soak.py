"""Soak of the one-launch-per-generation kernels: 2 x 10^6 generations in uneven pieces, each form against the same run on the two-launch kernels
(positions, log-pdfs, acceptance counters bit for bit; moments to rounding) and against the stationary variance of the target.
Usage (GPU box): python scripts/exp/generation_soak.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import kissmcmc_jl_amd as kmc

G = 2_000_000
for nw, nd in ((4096, 4), (4096, 8), (2048, 32)):
    th = np.random.default_rng(7).standard_normal((nw, nd))
    out = {}
    for label, dbg in (("one launch per generation", None), ("two launches", "fused=0")):
        if dbg: os.environ["KMC_DEBUG"] = dbg
        else: os.environ.pop("KMC_DEBUG", None)
        os.environ["KMC_LAUNCH"] = "graph"            # (the two-launch run on the table graph: no budget of parameter updates involved)
        t = time.perf_counter()
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 1000, 1, 2.0, 99, moments=True) as s:
            how = s.describe().split(",")[0]
            s.set_positions(th)
            left, rng = G, np.random.default_rng(3)
            while left > 0:
                n = int(min(left, rng.choice([1, 63, 64, 1000, 99999, 400001])))
                s.run(n); left -= n
            s.sync()
            msum, msq, cnt = s.moments()
            out[label] = (s.positions(), s.logp(), s.naccept(), msum / cnt, msq / cnt - (msum / cnt) ** 2, how)
        print(f"{nw} x {nd}: {label}: {time.perf_counter() - t:.1f} s wall, {how}; accept {out[label][2].mean() / (G - 1000):.4f}, "
              f"mean in [{out[label][3].min():+.4f}, {out[label][3].max():+.4f}], variance in [{out[label][4].min():.4f}, {out[label][4].max():.4f}]", flush=True)
    a, b = out["one launch per generation"], out["two launches"]
    same = np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.allclose(a[1], b[1], rtol=1e-12, atol=1e-12)
    print(f"  -> final positions and counters {'bit-identical' if same else 'DIFFERENT'}; moments agree to {np.max(np.abs(a[3] - b[3])):.1e} / {np.max(np.abs(a[4] - b[4])):.1e}", flush=True)
os.environ.pop("KMC_DEBUG", None); os.environ.pop("KMC_LAUNCH", None)
