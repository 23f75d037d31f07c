"""The copy-kernel by-walker stream under allocation churn: shapes alternate, big host arrays come and go between samplers (address
ranges recycled by the allocator between hipHostRegister / hipHostUnregister pairs).  See profiles/NOTES.md (one unexplained failure)."""
import os, sys
sys.path.insert(0, '.')
os.environ["KMC_CHAIN_BLOCK"] = "1"
os.environ["KMC_BYWALKER_COPY"] = "kernel"
import numpy as np
import kissmcmc_jl_amd as kmc
import oracle
shapes = [(1024, 7, 600, 50, 2), (512, 5, 400, 20, 1), (1500, 3, 300, 10, 1), (100, 2, 2000, 100, 3), (4096, 6, 200, 20, 1)]
refs = []
for (nw, nd, G, nburn, nthin) in shapes:
    th = np.random.default_rng(6).standard_normal((nw, nd))
    r = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, nthin, 2.0, 12, nthreads=8), th)
    refs.append((th, r["chain"].transpose(1, 0, 2).copy()))
rng = np.random.default_rng(0)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
keep = []
for it in range(N):
    k = int(rng.integers(0, len(shapes)))
    nw, nd, G, nburn, nthin = shapes[k]
    th, want = refs[k]
    # churn: allocate and free host arrays of assorted sizes, some kept alive for a while
    keep.append(np.full(int(rng.integers(1 << 10, 1 << 22)), 7.5))
    if len(keep) > 6: keep.pop(int(rng.integers(0, len(keep))))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, nthin, 2.0, 12, store_chain=True, store_logp=True, stream_chain=True, chain_by_walker=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        cw, lw = s.chain(by_walker=True)
    if not np.array_equal(cw, want):
        bad += 1
        b = np.flatnonzero(cw.ravel() != want.ravel())
        print(f"iteration {it} shape {shapes[k]}: {len(b)} elements differ, first at {b[0]}, last at {b[-1]}, base {cw.ctypes.data:#x}; stale marker 7.5 present: {(cw.ravel()[b] == 7.5).any()}", flush=True)
print(f"{bad} bad of {N}")
