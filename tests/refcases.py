"""The reference's own test cases (reference ``test/runtests.jl:52-107``; the two blob cases, :80-107,
are in tests/test_gpu_hostdensity.py) restated as data, for both the oracle pins and the GPU drop-in tests.  ``tol`` is the emcee
tolerance ``tole``; ``mstep`` the scale c of the Metropolis proposal ``theta -> c*randn(n) .+ theta`` and
``tolm`` the Metropolis tolerance (``test/metro.jl``)."""
import math

import numpy as np

E = math.e

# name, density key, params, theta0, niter, truths (mean, median, std, skewness), tol
CASES = [
    dict(name="normal(-5,3)", dens="gauss", params=[-5.0, 3.0], theta0=-4.0, niter=10 ** 4,
         mean=-5.0, median=-5.0, std=3.0, skew=0.0, tol=0.3, mstep=9.0, tolm=0.3),     # runtests.jl:53-56
    dict(name="lognormal(0,1)", dens="lognormal", params=[0.0, 1.0], theta0=0.4, niter=10 ** 7,
         mean=math.exp(0.5), median=1.0, std=math.sqrt((E - 1) * E),
         skew=(E + 2) * math.sqrt(E - 1), tol=0.3, mstep=7.5, tolm=0.4),                # runtests.jl:57-61
    dict(name="mvnormal2", dens="mvnormal2", params=dict(mean=[0.5, -0.25], cov=[[0.47, 1.8], [1.8, 7.0]]),
         theta0=[0.4, 0.3], niter=10 ** 5,
         mean=[0.5, -0.25], median=None, std=[math.sqrt(0.47), math.sqrt(7.0)], skew=None, tol=0.3,
         mstep=0.5, tolm=0.3),                                                          # runtests.jl:62-67
    dict(name="rosenbrock2", dens="rosen", params=[1.0, 100.0, 20.0], theta0=[0.0, 0.0], niter=10 ** 7,
         mean=[0.98, 10.3], median=None, std=[3.1, 13.8], skew=None, tol=0.6, mstep=0.5, tolm=0.6),   # runtests.jl:68-79
]
NWALKERS = 100        # runtests.jl:24
BALL_RADIUS = 0.1     # runtests.jl:23


def skewness(x):
    x = np.asarray(x, dtype=np.float64)
    m = x.mean()
    return float(((x - m) ** 3).mean() / ((x - m) ** 2).mean() ** 1.5)   # StatsBase.skewness


def check_mean_std(thetas, case, tol=None):
    """reference test/runtests.jl:36-43 (test_mean_std); tol defaults to the emcee tolerance `tole`."""
    tol = case["tol"] if tol is None else tol
    thetas = np.asarray(thetas, dtype=np.float64)
    std_ = np.asarray(case["std"], dtype=np.float64)
    mean = thetas.mean(axis=0)
    std = thetas.std(axis=0, ddof=1)
    assert np.all(np.abs(mean - np.asarray(case["mean"])) < np.abs(std_ * tol)), (case["name"], "mean", mean)
    assert np.all(np.abs(std - std_) < np.abs(std_ * tol)), (case["name"], "std", std)
    if case["median"] is not None:
        assert abs(np.median(thetas) - case["median"]) < abs(std_ * tol), (case["name"], "median")
    if case["skew"] is not None:
        assert abs(skewness(thetas) - case["skew"]) < abs(std_ * 2 * tol), (case["name"], "skew", skewness(thetas))
