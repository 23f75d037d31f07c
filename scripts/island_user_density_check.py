import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
for nd, S in ((32, 256), (16, 256), (32, 128)):
    out = {}
    for name, pdf in (("menu", kmc.GaussianIso()), ("expr", kmc.ExprDensity("-0.5*x*x"))):
        with kmc.Sampler(pdf, 8192, nd, 256, 64, 1, 2.0, 5, moments=True, island_gens=32, island_size=S) as s:
            s.set_positions(np.random.default_rng(1).standard_normal((8192, nd)))
            s.run(256); s.sync()
            out[name] = (s.positions(), s.naccept(), s.moments())
    same = np.array_equal(out["menu"][0], out["expr"][0]) and np.array_equal(out["menu"][1], out["expr"][1])
    v = out["expr"][2][1].sum() / out["expr"][2][2] / nd
    print(nd, S, "expr == menu:", same, "second moment", round(float(v), 4))
