import os, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
os.environ["KMC_CHAIN_BLOCK"] = "1"
os.environ["KMC_BYWALKER_COPY"] = sys.argv[1] if len(sys.argv) > 1 else "kernel"
import numpy as np
import kissmcmc_jl_amd as kmc
import oracle
nw, nd, G, nburn, nthin, seed = 1024, 7, 600, 50, 2, 12
th = np.random.default_rng(6).standard_normal((nw, nd))
ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, nthin, 2.0, seed, nthreads=8), th)
want = ref["chain"].transpose(1, 0, 2)
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
for it in range(N):
    junk = np.full((nw, 275, nd), 7.5)           # make stale host data recognisable
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, stream_chain=True, chain_by_walker=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        cw, lw = s.chain(by_walker=True)
        how = s.describe()
    if not np.array_equal(cw, want):
        bad += 1
        d = np.argwhere(np.any(cw != want, axis=2))
        ks = np.unique(d[:, 1]); ws = np.unique(d[:, 0])
        print(f"iteration {it}: {len(d)} rows differ; samples {ks[:12]} ... walkers {ws[:8]} .. {ws[-3:]} ({len(ws)} walkers)")
        k = ks[0]; w = d[d[:, 1] == k][0, 0]
        got = cw[w, k]
        # where does the wrong row come from?
        hits = np.argwhere(np.all(np.isclose(want[w], got), axis=1)).ravel()
        print("   wrong row equals this walker's sample(s):", hits[:5], "nan?", np.isnan(got).any(), "7.5?", (got == 7.5).all())
    del junk
print(f"{bad} bad of {N}; {how[:80]}")
