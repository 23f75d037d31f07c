"""Pins of the CPU oracle against everything the reference's own tests hold for the emcee path
(SURVEY.md §8c): the Random123 known-answer vectors for its Philox, the g-distribution
known answers (reference test/emcee.jl:2-14), and the statistical integration cases
(reference test/emcee.jl:17-48 over test/runtests.jl:52-79) with the reference's tolerances."""
import math

import numpy as np
import pytest

import refcases
from oracle import host as ohost


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    assert oracle.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert oracle.philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert oracle.philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def test_draws_are_in_range(oracle):
    for w in range(200):
        p, uz, ua = oracle.draw(7, 3, w, 50)
        assert 0 <= p < 50 and 0.0 < uz < 1.0 and 0.0 < ua < 1.0


def test_g_dist(oracle):
    """reference test/emcee.jl:2-14"""
    a = 3.5
    samples2 = np.array([oracle.sample_g(123, i, 0, a) for i in range(50000)])
    assert np.all((1 / a <= samples2) & (samples2 <= a))                       # :6
    assert math.isclose(oracle.cdf_g_inv(1, a), a, rel_tol=1e-8)               # :7  (Julia's ≈)
    assert math.isclose(oracle.cdf_g_inv(0, a), 1 / a, rel_tol=1e-8)           # :8
    z = np.arange(1 / a, a + 1e-12, 0.01)                                       # :9
    g = np.array([oracle.g_pdf(v, a) for v in z])
    meang = np.sum(z * g) * 0.01                                                # :10
    assert abs(samples2.mean() - meang) < 1e-2                                  # :11
    stdg = math.sqrt(np.sum((meang - z) ** 2 * g) * 0.01)                       # :12
    assert abs(samples2.std(ddof=1) - stdg) < 1e-2                              # :13
    # closed forms: mean (a+1+1/a)/3, second moment (a^2+a+1+1/a+1/a^2)/5
    assert abs(samples2.mean() - (a + 1 + 1 / a) / 3) < 1e-2


def _oracle_density(oracle, case):
    d = case["dens"]
    if d == "gauss":
        return oracle.GAUSSIAN_ISO, case["params"]
    if d == "lognormal":
        return oracle.LOGNORMAL, case["params"]
    if d == "rosen":
        return oracle.ROSENBROCK, case["params"]
    if d == "mvnormal2":
        P = np.linalg.inv(np.array(case["params"]["cov"]))
        m = case["params"]["mean"]
        return oracle.MVNORMAL2, [m[0], m[1], P[0, 0], 0.5 * (P[0, 1] + P[1, 0]), P[1, 1]]
    raise KeyError(d)


@pytest.mark.parametrize("case", refcases.CASES, ids=[c["name"] for c in refcases.CASES])
def test_reference_integration_cases(oracle, case):
    """reference test/emcee.jl:17-48 on the oracle."""
    did, params = _oracle_density(oracle, case)
    nw, niter = refcases.NWALKERS, case["niter"]
    rng = np.random.default_rng(2024)
    pdf = lambda th: oracle.logpdf(did, params, th)
    theta0s = ohost.make_theta0s(case["theta0"], refcases.BALL_RADIUS, pdf, nw, rng)   # emcee.jl:21
    assert len(theta0s) == nw
    ndim = int(np.size(case["theta0"]))
    G, nburn, ns = ohost.emcee_counts(niter, nw)
    cfg = oracle.make_config(did, params, nw, ndim, G, nburn, 1, 2.0, seed=99, nthreads=4)
    r = oracle.emcee(cfg, np.array(theta0s).reshape(nw, ndim), moments=False)
    assert r["status"] == 0
    chain = r["chain"]                                   # [sample][walker][dim]
    assert chain.shape == (niter // nw // 2, nw, ndim)   # emcee.jl:29,35
    thetas = chain.transpose(1, 0, 2)                    # thetas[w][k]
    t, acc, l, b = ohost.squash_walkers([list(map(tuple, w)) for w in thetas], list(r["accept_ratio"]),
                                        [list(w) for w in r["chain_logp"].T]) if niter <= 10 ** 5 else \
        (thetas.reshape(-1, ndim), float(r["accept_ratio"].mean()), r["chain_logp"].T.reshape(-1), None)
    t = np.asarray(t, dtype=np.float64).reshape(-1, ndim)
    assert len(t) == niter // 2 and len(l) == niter // 2  # emcee.jl:41-42
    assert acc > 0.1                                      # emcee.jl:43
    refcases.check_mean_std(t if ndim > 1 else t[:, 0], case)   # emcee.jl:44


def test_oracle_validation_follows_reference_asserts(oracle):
    mk = lambda **kw: oracle.make_config(oracle.GAUSSIAN_ISO, [0, 1], **{**dict(nwalkers=10, ndim=2, ngenerations=1), **kw})
    L = oracle.lib()
    import ctypes as C
    assert L.kmco_validate(C.byref(mk())) == oracle.OK
    assert L.kmco_validate(C.byref(mk(a_scale=1.0))) == oracle.ERR_A_SCALE            # samplers.jl:200
    assert L.kmco_validate(C.byref(mk(nwalkers=11))) == oracle.ERR_ODD_WALKERS        # :202
    assert L.kmco_validate(C.byref(mk(nwalkers=2))) == oracle.ERR_TOO_FEW_WALKERS     # :205
    assert L.kmco_validate(C.byref(mk(nwalkers=4))) == oracle.OK                      # nwalkers == ndim+2 is allowed


def test_analytic_moments_of_menu_densities(oracle):
    """Tier-1 truths: exponential (mean 1, var 1), isotropic Gaussian (mean 0, var 1)."""
    rng = np.random.default_rng(3)
    nw = 100
    th = 0.5 + 0.1 * np.abs(rng.standard_normal((nw, 1)))
    cfg = oracle.make_config(oracle.EXPONENTIAL, [1.0], nw, 1, 40000, 2000, 1, 2.0, seed=5, nthreads=4)
    r = oracle.emcee(cfg, th, store_chain=False)
    m = r["sum"] / r["nmoment"]
    v = r["sumsq"] / r["nmoment"] - m ** 2
    assert abs(m[0] - 1.0) < 0.03 and abs(v[0] - 1.0) < 0.06
    assert abs(r["accept_ratio"].mean() - 0.745) < 0.01      # SURVEY.md §6 anchor
    th = rng.standard_normal((256, 8))
    cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], 256, 8, 6000, 1000, 1, 2.0, seed=6, nthreads=4)
    r = oracle.emcee(cfg, th, store_chain=False)
    m = r["sum"] / r["nmoment"]
    v = r["sumsq"] / r["nmoment"] - m ** 2
    assert np.all(np.abs(m) < 0.05) and np.all(np.abs(v - 1.0) < 0.08)
