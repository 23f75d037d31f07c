"""GPU: the reference's own emcee test-suite (reference test/emcee.jl) run against the drop-in
host API (`emcee`, `make_theta0s`, `squash_walkers`) on the HIP path, plus the README sequence."""
import numpy as np
import pytest

import refcases

pytestmark = pytest.mark.gpu


def _pdf(kmc, case):
    d = case["dens"]
    if d == "gauss":
        return kmc.GaussianIso(*case["params"])
    if d == "lognormal":
        return kmc.LogNormal(*case["params"])
    if d == "rosen":
        return kmc.Rosenbrock(*case["params"])
    if d == "mvnormal2":
        return kmc.MvNormal2(case["params"]["mean"], case["params"]["cov"])
    raise KeyError(d)


@pytest.mark.parametrize("case", refcases.CASES, ids=[c["name"] for c in refcases.CASES])
def test_reference_emcee_testset(kmc, case):
    """reference test/emcee.jl:17-48, line by line."""
    pdf = _pdf(kmc, case)
    nw, niter = refcases.NWALKERS, case["niter"]
    theta0s = kmc.make_theta0s(case["theta0"], refcases.BALL_RADIUS, pdf, nw, rng=42)          # :21-23
    samples = kmc.emcee(pdf, theta0s, niter=niter, use_progress_meter=False, seed=4242)        # :24-28
    assert tuple(len(s) for s in samples[:3]) == (nw, nw, nw)                                  # :29
    assert samples[3] is None                                                                  # :33
    assert len(samples[0][0]) == niter // nw // 2                                              # :35
    thetas, accept_ratio, logdensities, blobs = kmc.squash_walkers(*samples, verbose=False)    # :36-38
    assert blobs is None                                                                       # :40
    assert len(thetas) == niter // 2                                                           # :41
    assert len(logdensities) == niter // 2                                                     # :42
    assert accept_ratio > 0.1                                                                  # :43
    refcases.check_mean_std(thetas, case)                                                      # :44


def test_readme_sequence(kmc, capsys):
    """reference README.md:15-27 with the menu density standing in for the closure."""
    logpdf = kmc.Exponential()                           # README.md:15  x<0 ? -Inf : -x
    theta0 = 0.5
    thetase, accept_ratioe, logd, blobs = kmc.emcee(logpdf, kmc.make_theta0s(theta0, 0.1, logpdf, 100, rng=1),
                                                    niter=10 ** 5, seed=2)           # README.md:25 (progress meter on)
    assert thetase.shape == (100, 500) and accept_ratioe.shape == (100,)
    assert "emcee, niter=100000, nwalkers=100" in capsys.readouterr().err
    t, acc = kmc.squash_walkers(thetase, accept_ratioe)[:2]                          # README.md:27
    assert t.shape == (50000,)
    assert abs(acc - 0.745) < 0.02                       # SURVEY.md §6 anchor
    assert abs(t.mean() - 1.0) < 0.08 and abs(t.var() - 1.0) < 0.15 and t.min() >= 0.0


def test_emcee_matches_oracle_through_the_public_api(kmc, oracle):
    """Same seeded inputs -> the drop-in API returns exactly the oracle's chains."""
    from oracle import host as ohost
    pdf = kmc.Rosenbrock()
    th = kmc.make_theta0s([0.0, 0.0], 0.1, pdf, 100, rng=9)
    thetas, acc, logd, _ = kmc.emcee(pdf, th, niter=40000, nthin=2, use_progress_meter=False, seed=31)
    G, nburn, ns = ohost.emcee_counts(40000, 100, None, 2)
    cfg = oracle.make_config(oracle.ROSENBROCK, [1.0, 100.0, 20.0], 100, 2, G, nburn, 2, 2.0, 31)
    r = oracle.emcee(cfg, th)
    assert thetas.shape == (100, ns, 2)
    np.testing.assert_array_equal(thetas, r["chain"].transpose(1, 0, 2))
    np.testing.assert_array_equal(acc, r["accept_ratio"])
    np.testing.assert_allclose(logd, r["chain_logp"].T, rtol=1e-12, atol=1e-12)


def test_input_is_not_mutated_and_nonfinite_start_is_rejected(kmc):
    pdf = kmc.Exponential()
    th = np.full(100, 0.5)
    before = th.copy()
    kmc.emcee(pdf, th, niter=1000, use_progress_meter=False, seed=1)
    np.testing.assert_array_equal(th, before)                         # samplers.jl:198
    th[7] = -1.0
    with pytest.raises(ValueError, match="non-finite initial log-pdf"):
        kmc.emcee(pdf, th, niter=1000, use_progress_meter=False, seed=1)


def test_zero_generations_and_degenerate_burnin(kmc):
    """niter < nwalkers -> niter_walker = 0 (samplers.jl:203): empty chains, accept_ratio = 0/0 like the reference."""
    pdf = kmc.GaussianIso()
    thetas, acc, logd, _ = kmc.emcee(pdf, np.zeros((10, 2)), niter=5, use_progress_meter=False, seed=1)
    assert thetas.shape == (10, 0, 2) and logd.shape == (10, 0)
    assert np.all(np.isnan(acc))


def test_readme_call_sequence_in_plain_c(tmp_path):
    """examples/readme_call.c: the reference README's emcee call (README.md:15-27) through the C ABI alone -- compiled
    with gcc against include/kissmcmc_hip.h and the shared library, no Python or torch in the process."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "readme_call")
    libdir = os.path.join(root, "kissmcmc.jl_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "readme_call.c"),
                           "-o", exe, "-L", libdir, "-lkissmcmc_hip", "-lm", f"-Wl,-rpath,{libdir}"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("samples 50000 ")


def test_walkers_of_matrix_shape(kmc):
    """The reference asks of a walker only `.+`, `.*` and `length` (src/samplers.jl:156): a 2 x 3 Matrix per walker works there
    (N = length(theta0s[1]) = 6, :243).  Here: theta0s [nwalkers, 2, 3] -- a device density sees the flattened walker, a host
    callable the shaped one -- and thetas come back [nwalkers, nsamples, 2, 3]; same chain as the flat call, bit for bit."""
    nw, shape = 64, (2, 3)
    th = np.random.default_rng(3).standard_normal((nw,) + shape)
    kw = dict(niter=nw * 40, use_progress_meter=False, seed=12)
    t_shaped, acc, lp, _ = kmc.emcee(kmc.GaussianIso(), th, **kw)
    t_flat, acc_f, lp_f, _ = kmc.emcee(kmc.GaussianIso(), th.reshape(nw, 6), **kw)
    assert t_shaped.shape == (nw, 20, 2, 3)
    np.testing.assert_array_equal(t_shaped.reshape(nw, 20, 6), t_flat)
    np.testing.assert_array_equal(acc, acc_f)
    seen = []

    def closure(x):                                   # a host callable receives the walker in ITS shape
        seen.append(np.shape(x))
        return -0.5 * float((x * x).sum())

    t_host, acc_h, lp_h, _ = kmc.emcee(closure, th, **kw)
    assert set(seen) == {shape}
    np.testing.assert_array_equal(t_host, t_shaped)          # same draws, same density values up to summation order -> same decisions
    flat, ar, _, _ = kmc.squash_walkers(t_shaped, acc, lp, verbose=False)
    assert flat.shape == (nw * 20, 2, 3)
