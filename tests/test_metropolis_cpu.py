"""CPU leg of the many-chain Metropolis path (reference src/samplers.jl:59-128): the oracle pinned on the
reference's own metropolis tests (reference test/metro.jl:2-21 over test/runtests.jl:52-79), on the
committed golden fixtures, and the host-side argument handling of the product API (no GPU needed)."""
import glob
import os

import numpy as np
import pytest

import refcases
from test_oracle_pins import _oracle_density

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metropolis")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load_golden(name):
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    for k in ("density", "nchains", "ndim", "niter", "nburnin", "nthin", "seed"):
        z[k] = int(z[k])
    return z


def compare_golden(z, r, exact):
    """exact: same libm as the generator (the oracle itself); else device libm vs glibc -> rounding-level tolerance."""
    np.testing.assert_array_equal(r["naccept"], z["naccept"])
    if exact:
        np.testing.assert_array_equal(r["final_pos"], z["final_pos"])
        np.testing.assert_array_equal(r["chain"][-1], z["chain_last"])
    tol = dict(rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(r["final_pos"], z["final_pos"], **tol)
    np.testing.assert_allclose(r["chain"][-1], z["chain_last"], **tol)
    np.testing.assert_allclose(r["final_logp"], z["final_logp"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(r["chain_logp"], z["chain_logp"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(r["chain_sum"], z["chain_sum"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(r["chain_sumsq"], z["chain_sumsq"], rtol=1e-10, atol=1e-10)


def test_metropolis_draws(oracle):
    """Box-Muller normals and the accept uniform of the seeded stream: range, moments, independence of ndim."""
    nrm = np.array([oracle.metropolis_draw(3, it, 0, 6)[0] for it in range(20000)])
    ua = np.array([oracle.metropolis_draw(3, it, 5, 1)[1] for it in range(20000)])
    assert np.all((0.0 < ua) & (ua < 1.0)) and abs(ua.mean() - 0.5) < 0.01
    assert np.all(np.abs(nrm.mean(axis=0)) < 0.03) and np.all(np.abs(nrm.std(axis=0) - 1.0) < 0.03)
    assert np.all(np.abs(np.corrcoef(nrm.T) - np.eye(6)) < 0.03)
    a, _ = oracle.metropolis_draw(9, 17, 4, 3)
    b, _ = oracle.metropolis_draw(9, 17, 4, 11)
    np.testing.assert_array_equal(a, b[:3])           # dimension d's normal does not depend on ndim


@pytest.mark.parametrize("case", refcases.CASES, ids=[c["name"] for c in refcases.CASES])
def test_reference_metropolis_cases(oracle, case):
    """reference test/metro.jl:2-21 on the oracle (one chain, as in the reference)."""
    did, params = _oracle_density(oracle, case)
    niter = case["niter"]
    th0 = np.atleast_1d(np.asarray(case["theta0"], dtype=np.float64))[None, :]
    r = oracle.metropolis(did, params, th0, case["mstep"], niter, seed=4242, moments=False)
    assert r["status"] == 0
    thetas, logd = r["chain"][:, 0, :], r["chain_logp"][:, 0]
    assert len(thetas) == niter // 2 and len(logd) == niter // 2                 # metro.jl:13-14
    assert 0.15 < r["accept_ratio"][0] < 0.45                                    # metro.jl:15
    refcases.check_mean_std(thetas if thetas.shape[1] > 1 else thetas[:, 0], case, case["tolm"])   # metro.jl:16


def test_metropolis_bookkeeping_follows_the_reference(oracle):
    """nsamples = (niter - nburnin) ÷ nthin (:88); stored states are the CURRENT state, moved or not (:113);
    counters restart after burn-in (:122-125); accept_ratio = naccept / (niter - nburnin) (:127)."""
    th0 = np.array([[0.3], [-0.2], [1.0]])
    r = oracle.metropolis(oracle.GAUSSIAN_ISO, [0.0, 1.0], th0, 2.5, 103, 40, 4, seed=8)
    assert r["chain"].shape == ((103 - 40) // 4, 3, 1)
    assert np.all(r["naccept"] <= 63) and np.all(r["accept_ratio"] == r["naccept"] / 63.0)
    # every stored log-density is the log-pdf of the stored state
    np.testing.assert_allclose(r["chain_logp"], -0.5 * r["chain"][:, :, 0] ** 2, rtol=0, atol=1e-15)
    # nburnin = 0: nothing is discarded, and a chain that never moves repeats its start
    r0 = oracle.metropolis(oracle.GAUSSIAN_ISO, [0.0, 1.0], th0, 0.0, 10, 0, 1, seed=8)
    assert r0["chain"].shape == (10, 3, 1)
    assert np.all(r0["chain"] == th0[None])           # step 0: the proposal equals the state; p1 - p0 = 0 > log u always
    assert np.all(r0["naccept"] == 10)
    # chains are independent streams: chain c of a 3-chain run equals a 1-chain run keyed... by its own index only
    r3 = oracle.metropolis(oracle.GAUSSIAN_ISO, [0.0, 1.0], th0[:2], 2.5, 103, 40, 4, seed=8)
    np.testing.assert_array_equal(r3["chain"], r["chain"][:, :2])


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_metropolis_golden(oracle, name):
    z = load_golden(name)
    r = oracle.metropolis(z["density"], list(z["params"]), z["theta0"], z["step"], z["niter"], z["nburnin"], z["nthin"], z["seed"])
    assert r["status"] == 0
    compare_golden(z, r, exact=False)      # tolerance, not bits: libm's log/sin/cos may differ between images


def test_threaded_metropolis_oracle_equals_serial(oracle):
    z = load_golden("gauss_96x7")
    a = oracle.metropolis(z["density"], list(z["params"]), z["theta0"], z["step"], z["niter"], z["nburnin"], z["nthin"], z["seed"], nthreads=1)
    b = oracle.metropolis(z["density"], list(z["params"]), z["theta0"], z["step"], z["niter"], z["nburnin"], z["nthin"], z["seed"], nthreads=4)
    np.testing.assert_array_equal(a["chain"], b["chain"])
    np.testing.assert_array_equal(a["naccept"], b["naccept"])


# ---- product host API (argument handling; compute needs the GPU) ------------------------------------------
def test_host_api_argument_handling(kmc):
    """Closures for pdf / sample_ppdf are accepted (host route; compute needs the GPU) -- what is refused is what the
    reference could not run either, or what cannot work by construction."""
    from kissmcmc_jl_amd import _lib
    with pytest.raises(TypeError, match="callable"):
        kmc.metropolis(kmc.GaussianIso(), "not a proposal", 0.0, niter=10)
    with pytest.raises(TypeError, match="callable"):
        kmc.metropolis(42, kmc.GaussianStep(1.0), 0.0, niter=10)
    with pytest.raises(NotImplementedError, match="returns a blob"):
        kmc.metropolis(kmc.GaussianIso(), kmc.GaussianStep(1.0), 0.0, niter=10, hasblob=True)      # a menu density returns the log-pdf alone
    with pytest.raises(ValueError, match="hasblob=True"):
        kmc.metropolis(kmc.GaussianIso(), kmc.GaussianStep(1.0), 0.0, niter=10, reduce_blob=lambda b, x: None)
    with pytest.raises(ValueError):
        kmc.metropolis_chains(kmc.Rosenbrock(), kmc.GaussianStep(0.5), np.zeros(8), niter=10)      # 1-D Rosenbrock
    with pytest.raises(ValueError, match="scales"):
        kmc.metropolis_chains(kmc.GaussianIso(), kmc.GaussianStep([1.0, 2.0, 3.0]), np.zeros((8, 2)), niter=10)
    if _lib.lib().kmc_device_count() == 0:
        # the README-style call with two closures reaches the library; without a GPU it fails there: no CPU fallback
        with pytest.raises(kmc.KmcError) as e:
            kmc.metropolis(lambda x: -x * x, lambda th: th + 1.0, 0.0, niter=10)
        assert e.value.status == _lib.ERR_NO_DEVICE


def test_c_abi_metropolis_validate(kmc):
    import ctypes as C
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    step = (C.c_double * 2)(0.5, 0.5)

    def cfg(**kw):
        c = _lib.MetropolisConfig()
        c.dtype, c.density = _lib.F64, _lib.GAUSSIAN_ISO
        c.params[0], c.params[1] = 0.0, 1.0
        c.nchains, c.ndim, c.niter, c.nburnin, c.nthin = 4, 2, 10, 5, 1
        c.step = C.cast(step, C.POINTER(C.c_double))
        for k, v in kw.items():
            setattr(c, k, v)
        return c
    assert L.kmc_metropolis_validate(C.byref(cfg())) == _lib.OK
    assert L.kmc_metropolis_validate(C.byref(cfg(nchains=0))) == _lib.ERR_BAD_ARG
    assert L.kmc_metropolis_validate(C.byref(cfg(nthin=0))) == _lib.ERR_BAD_ARG
    assert L.kmc_metropolis_validate(C.byref(cfg(step=None))) == _lib.ERR_BAD_ARG
    assert L.kmc_metropolis_validate(C.byref(cfg(density=_lib.ROSENBROCK, ndim=1))) == _lib.ERR_BAD_ARG
    assert L.kmc_metropolis_validate(C.byref(cfg(density=_lib.HOST_DENSITY))) == _lib.ERR_BAD_ARG       # no host_logpdf given
    cb = _lib.HOST_LOGPDF_FN(lambda rows, n, nd, out, user: 0)
    assert L.kmc_metropolis_validate(C.byref(cfg(density=_lib.HOST_DENSITY, host_logpdf=C.cast(cb, C.c_void_p)))) == _lib.OK
    assert L.kmc_metropolis_validate(C.byref(cfg(host_logpdf=C.cast(cb, C.c_void_p)))) == _lib.ERR_BAD_ARG   # needs KMC_HOST_DENSITY
    pb = _lib.HOST_PROPOSE_FN(lambda rows, n, nd, out, user: 0)
    assert L.kmc_metropolis_validate(C.byref(cfg(step=None, host_propose=C.cast(pb, C.c_void_p)))) == _lib.OK
    assert L.kmc_metropolis_validate(C.byref(cfg(flags=_lib.P2P))) == _lib.ERR_BAD_ARG
    assert L.kmc_metropolis_validate(None) == _lib.ERR_BAD_ARG
