"""Soak of the ragged two-launch kernels (tail chunks folded onto the row's last chunk, ndim in the pointer tag) in every launch mode: 30 000 generations in uneven pieces per shape and mode,
positions / counters / log-pdfs bit for bit across the modes, moments to 1e-10.
    python scripts/probes/ragged_soak.py   -> profiles/r05_ragged_soak.txt"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
rng = np.random.default_rng(7)
os.environ["KMC_DEBUG"] = "fused=0,updated-budget-mb=100000"
for name, pdf, nw, nd, scale in (("65536x31 gauss", kmc.GaussianIso, 65536, 31, 1.0), ("16384x63 rosen", kmc.Rosenbrock, 16384, 63, 0.1), ("8192x127 gauss", kmc.GaussianIso, 8192, 127, 1.0),
                                 ("262144x20 lognormal", kmc.LogNormal, 262144, 20, None), ("32768x50 gauss", kmc.GaussianIso, 32768, 50, 1.0), ("4096x201 gauss", kmc.GaussianIso, 4096, 201, 1.0)):
    G = 30000
    th = np.exp(0.3 * rng.standard_normal((nw, nd))) if scale is None else scale * rng.standard_normal((nw, nd))
    out = {}
    for mode in ("graph", "updated", "eager"):
        os.environ["KMC_LAUNCH"] = mode
        with kmc.Sampler(pdf(), nw, nd, G, G // 4, 11, 2.0, 99, moments=True) as s:
            assert "ragged" in s.describe(), s.describe()
            s.set_positions(th)
            done = 0
            for p in rng.integers(1, 3000, size=60):
                p = int(min(p, G - done))
                if p <= 0: break
                s.run(p); done += p
            if done < G: s.run(G - done)
            s.sync()
            out[mode] = (s.positions(), s.naccept(), s.logp(), s.moments())
    ok = all(np.array_equal(out[m][k], out["graph"][k]) for m in ("updated", "eager") for k in (0, 1, 2))
    mom = all(np.allclose(out[m][3][0], out["graph"][3][0], rtol=1e-10, atol=1e-6) and out[m][3][2] == out["graph"][3][2] for m in ("updated", "eager"))
    print(f"{name}: {G} generations in uneven pieces, table graph | updated graph | eager launches: positions, counters, log-pdfs {'bit-identical' if ok else 'DIFFER'}, moments {'equal to 1e-10' if mom else 'DIFFER'} "
          f"(accept {out['graph'][1].sum() / nw / (G - G // 4):.3f})", flush=True)
