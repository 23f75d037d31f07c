"""Integrated autocorrelation time (reference src/analysis.jl:140-167, :252-285 -- commented-out code, followed as
written): the numpy restatement pinned on processes with known tau, and the GPU implementation against it."""
import numpy as np
import pytest

from oracle import host as ohost


def ar1(phi, nsamples, nchains, ntheta=1, seed=0):
    rng = np.random.default_rng(seed)
    x = np.zeros((ntheta, nsamples, nchains))
    e = rng.standard_normal((ntheta, nsamples, nchains))
    x[:, 0] = e[:, 0] / np.sqrt(1 - phi * phi)
    for t in range(1, nsamples):
        x[:, t] = phi * x[:, t - 1] + e[:, t]
    return x


def test_oracle_int_acorr_known_answers():
    """AR(1) with coefficient phi has tau = (1 + phi) / (1 - phi); white noise has tau = 1."""
    for phi in (0.5, 0.8, 0.9):
        tau, conv = ohost.int_acorr(ar1(phi, 6000, 48, seed=3))
        assert abs(tau[0] - (1 + phi) / (1 - phi)) < 0.08 * (1 + phi) / (1 - phi)
        assert np.allclose(conv, 6000 / tau)
    tau, _ = ohost.int_acorr(np.random.default_rng(1).standard_normal((3, 2001, 32)))
    assert np.all(np.abs(tau - 1.0) < 0.1)
    # acor1d: first half of the circular autocorrelation, lag 0 normalised to 1 (:264-267)
    a = ohost.acor1d(np.arange(11.0))
    assert len(a) == 5 and a[0] == 1.0
    assert ohost.auto_window(np.array([3.0, 2.0, 1.5, 0.7, 0.5]), 5) == 3     # first i (1-based 4) with i >= 5 * 0.7
    neff, thin, conv, ns, tau2, convs = ohost.eff_samples(ar1(0.8, 4000, 16, seed=5))
    assert abs(neff - 4000 * 16 / tau2[0]) <= 1 and thin in (8, 9, 10)


def test_host_api_checks(kmc):
    with pytest.raises(AssertionError):
        kmc.int_acorr(np.zeros((4, 100)), c=1.0)
    with pytest.raises(ValueError):
        kmc.int_acorr(np.zeros(100))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(48, 1000, 1), (16, 777, 5), (1, 64, 2), (300, 129, 32)])
def test_gpu_int_acorr_equals_the_restatement(kmc, shape):
    """kmc_int_acorr (hipFFT + kernels) against numpy, emcee output layout [walker][sample][dim]; odd lengths too."""
    nw, ns, nd = shape
    x = ar1(0.7, ns, nw, ntheta=nd, seed=nd)                   # (ntheta, nsamples, nchains)
    thetas = np.ascontiguousarray(x.transpose(2, 1, 0))        # [walker][sample][dim]
    tau, conv = kmc.int_acorr(thetas, warn=False)
    rtau, rconv = ohost.int_acorr(x)
    np.testing.assert_allclose(tau, rtau, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(conv, rconv, rtol=1e-9, atol=1e-9)
    if nd == 1:
        t1, _ = kmc.int_acorr(thetas[:, :, 0], warn=False)     # scalar walkers
        np.testing.assert_allclose(t1, rtau, rtol=1e-9)


@pytest.mark.gpu
def test_gpu_diagnostics_on_a_real_run(kmc):
    """README.md:26 of the reference: "check convergence using integrated autocorrelation" -- on an emcee run."""
    pdf = kmc.GaussianIso()
    th0 = kmc.make_theta0s(np.zeros(4), 0.1, pdf, 128, rng=0)
    thetas, acc, logd, _ = kmc.emcee(pdf, th0, niter=128 * 4000, use_progress_meter=False, seed=3)
    tau, conv = kmc.int_acorr(thetas, warn=False)
    assert thetas.shape == (128, 2000, 4) and tau.shape == (4,)
    assert np.all((tau > 2) & (tau < 40)) and np.all(conv > 50)
    neff, thin, mconv, ns, taus, convs = kmc.eff_samples(thetas)
    assert 128 * 2000 / 40 < neff < 128 * 2000 / 2 and np.allclose(taus, tau)
    with pytest.warns(UserWarning, match="likely not accurate"):
        kmc.int_acorr(thetas[:, :60], warn=True)


@pytest.mark.gpu
def test_gpu_sampler_int_acorr_runs_on_the_device_chain(kmc):
    nw, nd, G = 256, 8, 1200
    th = np.random.default_rng(2).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 200, 1, 2.0, 5, store_chain=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        tau, conv = s.int_acorr()
        chain, _ = s.chain(logp=False)
    rtau, rconv = kmc.int_acorr(chain.transpose(1, 0, 2), warn=False)
    np.testing.assert_allclose(tau, rtau, rtol=1e-12)
    otau, _ = ohost.int_acorr(chain.transpose(2, 0, 1))
    np.testing.assert_allclose(tau, otau, rtol=1e-9)
