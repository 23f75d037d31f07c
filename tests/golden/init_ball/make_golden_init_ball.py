"""Generates tests/golden/init_ball/*.npz with the CPU oracle's seeded initial ball (oracle/kmc_oracle.c:
kmco_init_ball, the restatement of reference src/samplers.jl:311-349 the device-side kmc_sampler_init_ball is
checked against).  Fixtures hold inputs and expected outputs only.
Re-run:  python tests/golden/init_ball/make_golden_init_ball.py [--all]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import oracle  # noqa: E402

# name, density, params, theta0, radius, nwalkers, ndim, halving_steps, ntries, seed
CASES = [
    ("gauss_300x32", oracle.GAUSSIAN_ISO, [0.0, 1.0], "lin", 0.1, 300, 32, 7, 100, 42),          # every first try admissible
    ("expo_retry_256x3", oracle.EXPONENTIAL, [1.0], 0.02, 0.1, 256, 3, 7, 100, 7),                # ~70 % of first tries fail
    ("expo_shrink_200x8", oracle.EXPONENTIAL, [1.0], 0.01, 1.0, 200, 8, 7, 3, 5),                 # ball shrinks 1, 1/2, 1/8, ... per walker
    ("lognormal_odd_150x5", oracle.LOGNORMAL, [0.0, 1.0], 0.3, 0.5, 150, 5, 7, 100, 9),           # odd ndim (padded rows on the device)
    ("expo_readme_100x1", oracle.EXPONENTIAL, [1.0], 0.5, 0.1, 100, 1, 7, 100, 3),                # README.md:25 make_theta0s(0.5, 0.1, logpdf, 100)
    ("expo_fail_64x2", oracle.EXPONENTIAL, [1.0], -50.0, 0.1, 64, 2, 3, 4, 1),                    # no admissible point: every walker fails
]


def theta0_of(kind, nd):
    return np.linspace(-1.0, 1.0, nd) if isinstance(kind, str) else np.full(nd, float(kind))


def main():
    for name, did, params, t0, rad, nw, nd, hs, nt, seed in CASES:
        path = os.path.join(HERE, name + ".npz")
        if os.path.exists(path) and "--all" not in sys.argv:
            continue
        th = theta0_of(t0, nd)
        r = oracle.init_ball(did, params, th, rad, nw, nd, seed=seed, halving_steps=hs, ntries=nt)
        np.savez_compressed(path, density=did, params=np.array(params, dtype=np.float64), theta0=th, radius=np.full(nd, float(rad)),
                            nwalkers=nw, ndim=nd, halving_steps=hs, ntries=nt, seed=seed,
                            pos=r["pos"], logp=r["logp"], attempts=r["attempts"], nfail=r["nfail"])
        a = r["attempts"]
        print(name, "nfail", r["nfail"], "tries used: max", a.max(), "walkers needing > 1:", int((a > 1).sum()))


if __name__ == "__main__":
    main()
