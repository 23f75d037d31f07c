"""Soak of the one-launch-per-generation kernels at the 4-8 MiB states round 5 opened to them (launch mode as measured: table or updated graph): 2 x 10^5 generations in uneven pieces against the two-launch kernels
(KMC_DEBUG=fused=0) -- positions and counters bit for bit, moments to 1e-10.   python scripts/generation_soak.py   -> profiles/r05_generation_soak.txt"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
rng = np.random.default_rng(1)
for name, pdf, nw, nd, scale in (("8192x64 gauss", kmc.GaussianIso, 8192, 64, 1.0), ("16384x32 gauss", kmc.GaussianIso, 16384, 32, 1.0), ("8192x60 rosen", kmc.Rosenbrock, 8192, 60, 0.1), ("32768x16 expo", kmc.Exponential, 32768, 16, None),
                                  ("16384x64 rosen (C3)", kmc.Rosenbrock, 16384, 64, 0.1), ("32768x31 gauss", kmc.GaussianIso, 32768, 31, 1.0), ("4096x4 gauss", kmc.GaussianIso, 4096, 4, 1.0)):      # (8 MiB states, a ragged one and generation_lane: the second half of round 5)
    G = 200000
    th = (0.5 + 0.1 * np.abs(rng.standard_normal((nw, nd)))) if scale is None else scale * rng.standard_normal((nw, nd))
    out = {}
    for label, dbg in (("one", None), ("two", "fused=0")):
        if dbg: os.environ["KMC_DEBUG"] = dbg
        else: os.environ.pop("KMC_DEBUG", None)
        with kmc.Sampler(pdf(), nw, nd, G, G // 4, 7, 2.0, 99, moments=True) as s:
            assert ("one launch per generation" in s.describe()) == (label == "one"), s.describe()
            s.set_positions(th)
            done = 0
            pieces = rng.integers(1, 5000, size=400) if label == "one" else [G]
            for p in pieces:
                p = int(min(p, G - done))
                if p <= 0: break
                s.run(p); done += p
            if done < G: s.run(G - done)
            s.sync()
            out[label] = (s.positions(), s.naccept(), s.logp(), s.moments())
    same = np.array_equal(out["one"][0], out["two"][0]) and np.array_equal(out["one"][1], out["two"][1])
    m1, m2 = out["one"][3], out["two"][3]
    mom = np.allclose(m1[0], m2[0], rtol=1e-10, atol=1e-6) and np.allclose(m1[1], m2[1], rtol=1e-10, atol=1e-6) and m1[2] == m2[2]
    print(f"{name}: {G} generations in uneven pieces, one launch per generation vs two-launch kernels: positions and counters {'bit-identical' if same else 'DIFFER'}, moments {'equal to 1e-10' if mom else 'DIFFER'} (n = {m1[2]})", flush=True)
