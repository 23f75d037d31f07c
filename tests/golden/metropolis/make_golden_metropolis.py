"""Generates tests/golden/metropolis/*.npz with the CPU oracle's many-chain Metropolis
(oracle/kmc_oracle.c: kmco_metropolis) in this container: inputs + expected outputs only, no code.
The reference (Julia) cannot run here and holds no golden vectors.
Re-run:  python tests/golden/metropolis/make_golden_metropolis.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import oracle  # noqa: E402

# name, density id, params, nchains, ndim, step, niter, nburnin, nthin, seed, init
CASES = [
    ("normal_300x1", oracle.GAUSSIAN_ISO, [-5.0, 3.0], 300, 1, 9.0, 120, 60, 1, 21, "shift"),          # runtests.jl:53-56
    ("mvnormal_256x2", oracle.MVNORMAL2, [0.5, -0.25, 1.0 / 0.47, 0.0, 1.0 / 7.0], 256, 2, 0.5, 90, 30, 2, 22, "small"),
    ("rosen_128x2", oracle.ROSENBROCK, [1.0, 100.0, 20.0], 128, 2, 0.5, 150, 50, 1, 23, "small"),      # runtests.jl:68-79
    ("lognormal_200x1", oracle.LOGNORMAL, [0.0, 1.0], 200, 1, 7.5, 100, 40, 3, 24, "positive"),         # runtests.jl:57-61
    ("gauss_96x7", oracle.GAUSSIAN_ISO, [0.0, 1.0], 96, 7, [0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9], 80, 20, 1, 25, "normal"),
    ("expo_64x20", oracle.EXPONENTIAL, [1.0], 64, 20, 0.05, 60, 20, 1, 26, "positive"),                 # > 16 dims: chain kept in memory
]


def theta0(kind, nc, nd, seed):
    rng = np.random.default_rng(seed)
    if kind == "normal":
        return rng.standard_normal((nc, nd))
    if kind == "shift":
        return -4.0 + 0.1 * rng.standard_normal((nc, nd))
    if kind == "positive":
        return 0.55 + 0.1 * np.abs(rng.standard_normal((nc, nd)))
    if kind == "small":
        return 0.1 * rng.standard_normal((nc, nd))
    raise KeyError(kind)


def main():
    for name, did, params, nc, nd, step, niter, nburn, nthin, seed, init in CASES:
        th = theta0(init, nc, nd, seed)
        r = oracle.metropolis(did, params, th, step, niter, nburn, nthin, seed)
        assert r["status"] == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), density=did, params=np.array(params, dtype=np.float64),
                            nchains=nc, ndim=nd, step=np.broadcast_to(np.asarray(step, dtype=np.float64), (nd,)).copy(),
                            niter=niter, nburnin=nburn, nthin=nthin, seed=seed, theta0=th,
                            final_pos=r["final_pos"], final_logp=r["final_logp"], naccept=r["naccept"],
                            chain_last=r["chain"][-1], chain_logp=r["chain_logp"], chain_sum=r["chain_sum"],
                            chain_sumsq=r["chain_sumsq"])
        print(name, "accept", r["accept_ratio"].mean())


if __name__ == "__main__":
    main()
