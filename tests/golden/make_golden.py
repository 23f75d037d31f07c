"""Generates tests/golden/*.npz with the CPU oracle (oracle/kmc_oracle.c) in this container.

The reference (Julia) cannot run here and holds no golden vectors of its own,
so these fixtures pin the *oracle's* seeded output: inputs + expected outputs only, no code.
Re-run:  python tests/golden/make_golden.py [--all]   (without --all only missing fixtures are written)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

# name, density id, params, nwalkers, ndim, G, nburnin, nthin, a, seed, init
CASES = [
    ("gauss_64x4", oracle.GAUSSIAN_ISO, [0.0, 1.0], 64, 4, 50, 10, 1, 2.0, 11, "normal"),
    ("gauss_256x32", oracle.GAUSSIAN_ISO, [0.0, 1.0], 256, 32, 40, 10, 2, 2.0, 12, "normal"),
    ("gauss_shift_100x1", oracle.GAUSSIAN_ISO, [-5.0, 3.0], 100, 1, 120, 60, 1, 2.0, 13, "shift"),
    ("expo_100x1_readme", oracle.EXPONENTIAL, [1.0], 100, 1, 200, 100, 1, 2.0, 14, "positive"),
    ("expo_128x16", oracle.EXPONENTIAL, [1.0], 128, 16, 40, 10, 3, 2.0, 15, "positive"),
    ("rosen_100x2", oracle.ROSENBROCK, [1.0, 100.0, 20.0], 100, 2, 200, 100, 1, 2.0, 16, "small"),
    ("rosen_256x64", oracle.ROSENBROCK, [1.0, 100.0, 20.0], 256, 64, 30, 10, 1, 2.0, 17, "small"),
    ("gauss_a35_128x8", oracle.GAUSSIAN_ISO, [0.0, 1.0], 128, 8, 60, 0, 1, 3.5, 18, "normal"),
    ("gauss_1040x1024", oracle.GAUSSIAN_ISO, [0.0, 1.0], 1040, 1024, 3, 1, 1, 2.0, 19, "hash"),
    # float rows (KMC_F32 / the oracle's state_f32): names end in _f32
    ("gauss_256x32_f32", oracle.GAUSSIAN_ISO, [0.0, 1.0], 256, 32, 40, 10, 2, 2.0, 21, "normal"),
    ("rosen_256x64_f32", oracle.ROSENBROCK, [1.0, 100.0, 20.0], 256, 64, 30, 10, 1, 2.0, 22, "small"),
    ("expo_100x1_f32", oracle.EXPONENTIAL, [1.0], 100, 1, 200, 100, 1, 2.0, 23, "positive"),
]


def theta0(kind, nw, nd, seed):
    if kind == "hash":   # exact, formula-defined input for the big case (not stored in the fixture)
        w = np.arange(nw, dtype=np.uint64)[:, None]
        d = np.arange(nd, dtype=np.uint64)[None, :]
        k = (w * np.uint64(1315423911) + d * np.uint64(2654435761) + np.uint64(seed)) % np.uint64(1 << 20)
        return k.astype(np.float64) / float(1 << 19) - 1.0
    rng = np.random.default_rng(seed)
    if kind == "normal":
        return rng.standard_normal((nw, nd))
    if kind == "shift":
        return -4.0 + 0.1 * rng.standard_normal((nw, nd))
    if kind == "positive":
        return 0.55 + 0.1 * np.abs(rng.standard_normal((nw, nd)))
    if kind == "small":
        return 0.1 * rng.standard_normal((nw, nd))
    raise KeyError(kind)


def main():
    for name, did, params, nw, nd, G, nburn, nthin, a, seed, init in CASES:
        if os.path.exists(os.path.join(HERE, name + ".npz")) and "--all" not in sys.argv:
            continue                                   # existing fixtures are rewritten only on request
        th = theta0(init, nw, nd, seed)
        cfg = oracle.make_config(did, params, nw, nd, G, nburn, nthin, a, seed, state_f32=name.endswith("_f32"))
        r = oracle.emcee(cfg, th)
        assert r["status"] == 0
        big = init == "hash"
        out = dict(density=did, params=np.array(params, dtype=np.float64), nwalkers=nw, ndim=nd, G=G,
                   nburnin=nburn, nthin=nthin, a_scale=a, seed=seed, init=init,
                   final_logp=r["final_logp"], naccept=r["naccept"], sum=r["sum"], sumsq=r["sumsq"],
                   nmoment=r["nmoment"])
        if big:   # keep the fixture small: formula-defined input, row checksums of the output
            out.update(final_pos_rowsum=r["final_pos"].sum(axis=1), final_pos_head=r["final_pos"][:4])
        else:
            out.update(theta0=th, final_pos=r["final_pos"], chain_last=r["chain"][-1], chain_logp=r["chain_logp"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "accept", r["accept_ratio"].mean())


if __name__ == "__main__":
    main()
