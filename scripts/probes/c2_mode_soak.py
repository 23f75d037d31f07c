"""C2 for 10^6 generations in each launch mode (table graph, updated graph with replays of 128 generations, eager launches), cut into uneven pieces: the three runs must end in the same
positions, log-pdfs, counters and moments bit for bit (round 2's soak, repeated on the final tree of round 5).   python scripts/probes/c2_mode_soak.py [generations]   -> profiles/r05_c2_mode_soak.txt"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
os.environ["KMC_DEBUG"] = "updated-budget-mb=100000"
G = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 6
nw, nd = 65536, 32
rng = np.random.default_rng(3)
th = rng.standard_normal((nw, nd))
pieces = [int(p) for p in rng.integers(1, 40000, size=200)]
out = {}
for mode in ("graph", "updated", "eager"):
    os.environ["KMC_LAUNCH"] = mode
    t0 = time.time()
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, 1, 2.0, 11, moments=True) as s:
        s.set_positions(th)
        done = 0
        for p in pieces:
            p = min(p, G - done)
            if p <= 0: break
            s.run(p); done += p
        if done < G: s.run(G - done)
        s.sync()
        out[mode] = (s.positions(), s.logp(), s.naccept(), s.moments())
        how = s.describe().split(", grid")[1][:110]
    print(f"KMC_LAUNCH={mode}: {G} generations ({2 * G} launches) in {time.time() - t0:.1f} s;{how}", flush=True)
same = all(np.array_equal(out[m][k], out["graph"][k]) for m in ("updated", "eager") for k in (0, 1, 2))
mom = all(np.array_equal(out[m][3][0], out["graph"][3][0]) and np.array_equal(out[m][3][1], out["graph"][3][1]) and out[m][3][2] == out["graph"][3][2] for m in ("updated", "eager"))
msum, msq, n = out["graph"][3]
var = msq / n - (msum / n) ** 2
print(f"positions, log-pdfs, counters: {'bit-identical' if same else 'DIFFER'} across the three modes; moments: {'bit-identical' if mom else 'DIFFER'}; accept {out['graph'][2].sum() / nw / (G - G // 2):.4f}; "
      f"variance of every dimension within {np.abs(var - 1).max():.1e} of 1 (n = {n})", flush=True)
