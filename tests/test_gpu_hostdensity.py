"""GPU: host-evaluated log-densities (HostLogPdf / KMC_HOST_DENSITY) -- the reference's arbitrary `pdf`
closure (src/samplers.jl:257) kept on the host while the stretch move, the draws, the accept test,
the counters and the storage run on the device.  When the closure is the oracle's own density the
whole run must equal the oracle's bit for bit (log-pdfs included: they ARE the closure's values)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(kmc, pdf, th, G, nburn, nthin, seed):
    nw, nd = th.shape
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
        assert "host-evaluated" in s.describe()
        s.set_positions(th)
        s.run(G // 2)
        s.run(G - G // 2)
        s.sync()
        ch, cl = s.chain()
        m = s.moments()
        return dict(pos=s.positions(), logp=s.logp(), nacc=s.naccept(), chain=ch, chain_logp=cl, sum=m[0], sumsq=m[1],
                    n=m[2], acc=s.accept_ratio())


CASES = [
    ("exponential", 1, [1.0], 100, 1, 120, 60, 1),      # the README shape
    ("gaussian", 0, [0.0, 1.0], 64, 5, 50, 10, 3),
    ("rosenbrock", 2, [1.0, 100.0, 20.0], 40, 3, 40, 7, 2),
    ("gaussian-wide", 0, [-5.0, 3.0], 2100, 1030, 3, 1, 1),   # beyond every vector plan
]


@pytest.mark.parametrize("name,did,params,nw,nd,G,nburn,nthin", CASES)
def test_oracle_density_as_host_closure_is_bit_identical(kmc, oracle, name, did, params, nw, nd, G, nburn, nthin):
    rng = np.random.default_rng(nw + nd)
    th = 0.5 + 0.1 * np.abs(rng.standard_normal((nw, nd))) if did == 1 else 0.3 * rng.standard_normal((nw, nd))
    pdf = kmc.HostLogPdf(lambda x: oracle.logpdf(did, params, x))
    got = _run(kmc, pdf, th, G, nburn, nthin, 17)
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, nthin, 2.0, 17), th)
    np.testing.assert_array_equal(got["nacc"], ref["naccept"])
    np.testing.assert_array_equal(got["pos"], ref["final_pos"])
    np.testing.assert_array_equal(got["logp"], ref["final_logp"])
    np.testing.assert_array_equal(got["chain"], ref["chain"])
    np.testing.assert_array_equal(got["chain_logp"], ref["chain_logp"])
    np.testing.assert_array_equal(got["acc"], ref["accept_ratio"])
    assert got["n"] == ref["nmoment"]
    np.testing.assert_allclose(got["sum"], ref["sum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(got["sumsq"], ref["sumsq"], rtol=1e-11, atol=1e-9)


def test_vectorized_closure(kmc, oracle):
    th = 0.3 * np.random.default_rng(2).standard_normal((128, 6))
    calls = []

    def batch(X):
        calls.append(X.shape)
        return np.array([oracle.logpdf(0, [0.0, 1.0], r) for r in X])

    got = _run(kmc, kmc.HostLogPdf(batch, vectorized=True), th, 20, 0, 1, 5)
    ref = oracle.emcee(oracle.make_config(0, [0.0, 1.0], 128, 6, 20, 0, 1, 2.0, 5), th)
    np.testing.assert_array_equal(got["chain"], ref["chain"])
    assert calls[0] == (128, 6) and set(calls[1:]) == {(64, 6)} and len(calls) == 1 + 2 * 20


def test_emcee_with_a_plain_closure_like_the_reference_tests(kmc):
    """reference test/runtests.jl:80-86: pdf = x -> -(x+5)^2/(2*3^2), scalar walkers, tol 0.3 sigma."""
    lp = lambda x: -(x + 5) ** 2 / (2 * 3 ** 2)
    theta0s = kmc.make_theta0s(-5.0, 0.1, lp, 100, rng=1)
    thetas, accept_ratio, logdensities, blobs = kmc.emcee(lp, theta0s, niter=10 ** 5, use_progress_meter=False, seed=3)
    assert thetas.shape == (100, 500) and logdensities.shape == (100, 500) and blobs is None
    assert np.all(accept_ratio > 0.1)
    t, ar, l, _ = kmc.squash_walkers(thetas, accept_ratio, logdensities, verbose=False)
    assert abs(t.mean() + 5) < 0.3 * 3 and abs(t.std() - 3) < 0.3 * 3
    np.testing.assert_array_equal(l, [lp(float(v)) for v in t])   # stored log-densities are the closure's values


def test_closure_exceptions_surface_in_python(kmc):
    th = np.random.default_rng(0).standard_normal((16, 2))
    n = [0]

    def bad(x):
        n[0] += 1
        if n[0] > 40:
            raise ZeroDivisionError("boom")
        return -0.5 * float(x @ x)

    with kmc.Sampler(kmc.HostLogPdf(bad), 16, 2, 10) as s:
        s.set_positions(th)
        with pytest.raises(ZeroDivisionError, match="boom"):
            s.run(10)
    with pytest.raises(ValueError, match="non-finite initial"):
        kmc.emcee(lambda x: -np.inf, np.zeros((8, 2)), niter=80, use_progress_meter=False)


def test_host_density_through_the_c_abi_one_shot(kmc, oracle):
    """kmc_emcee_run with kmc_config.host_logpdf: what a Julia @cfunction / C caller binds."""
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    nw, nd, G = 32, 4, 25
    th = np.ascontiguousarray(0.3 * np.random.default_rng(9).standard_normal((nw, nd)))

    @_lib.HOST_LOGPDF_FN
    def cb(rows, nrows, ndim, out, user):
        X = np.ctypeslib.as_array(rows, shape=(nrows, ndim))
        for i in range(nrows):
            out[i] = oracle.logpdf(0, [0.0, 1.0], X[i])
        return 0

    c = _lib.Config()
    c.dtype, c.density = _lib.F64, _lib.HOST_DENSITY
    c.nwalkers, c.ndim, c.ngenerations, c.nburnin, c.nthin, c.a_scale, c.seed = nw, nd, G, 5, 1, 2.0, 4
    c.flags = _lib.STORE_CHAIN
    c.host_logpdf = C.cast(cb, C.c_void_p)
    chain = np.zeros((G - 5, nw, nd))
    acc = np.zeros(nw)
    out = _lib.Outputs()
    dp = C.POINTER(C.c_double)
    out.chain, out.accept_ratio = chain.ctypes.data_as(dp), acc.ctypes.data_as(dp)
    _lib.check(L.kmc_emcee_run(C.byref(c), th.ctypes.data_as(dp), C.byref(out)))
    ref = oracle.emcee(oracle.make_config(0, [0.0, 1.0], nw, nd, G, 5, 1, 2.0, 4), th)
    np.testing.assert_array_equal(chain, ref["chain"])
    np.testing.assert_array_equal(acc, ref["accept_ratio"])
    # a missing callback and unsupported combinations are refused up front
    c.host_logpdf = None
    assert L.kmc_validate(C.byref(c)) == _lib.ERR_BAD_ARG
    c.host_logpdf = C.cast(cb, C.c_void_p)
    c.flags = _lib.P2P
    assert L.kmc_validate(C.byref(c)) == _lib.ERR_UNSUPPORTED


# ---- blobs (hasblob=true): src/samplers.jl:150-151, :208-210, :264, :270; test/runtests.jl:80-107 ----

def test_blobs_follow_the_walkers_exactly(kmc):
    """blob = (theta, p): the stored blobs must then equal the stored thetas / logdensities entry by entry --
    blob0s[nc] = blob1 exactly when theta0s[nc] = theta1 (:261-264), reduce_blob! exactly when stored (:268-271)."""
    lp = lambda x: -0.5 * float(x @ x)
    th = 0.4 * np.random.default_rng(11).standard_normal((24, 3))
    kw = dict(niter=24 * 60, nburnin=24 * 13, nthin=3, use_progress_meter=False, seed=21)
    thetas, acc, logd, blobs = kmc.emcee(lambda x: (lp(x), (x.copy(), lp(x))), th, hasblob=True, **kw)
    assert len(blobs) == 24 and all(len(b) == thetas.shape[1] == (60 - 13) // 3 for b in blobs)
    for w in range(24):
        np.testing.assert_array_equal(np.array([b[0] for b in blobs[w]]), thetas[w])
        np.testing.assert_array_equal(np.array([b[1] for b in blobs[w]]), logd[w])
    # and the sampler itself is unchanged by carrying blobs
    t2, a2, l2, b2 = kmc.emcee(lp, th, **kw)
    assert b2 is None
    np.testing.assert_array_equal(t2, thetas)
    np.testing.assert_array_equal(a2, acc)
    # squash_walkers: default append! / order=true keep blobs aligned with the samples (:408-421)
    for order in (False, True):
        t, _, l, b = kmc.squash_walkers(thetas, acc, logd, blobs, order=order, verbose=False)
        np.testing.assert_array_equal(np.array([x[0] for x in b]), t)
        np.testing.assert_array_equal(np.array([x[1] for x in b]), l)


def test_reference_blob_cases(kmc):
    """reference test/runtests.jl:80-107 through test/emcee.jl:21-45: blob = ones(1000), default reductions
    (blob_truths = 5000 x ones(1000)) and the sum reduction (blob_truths = [5000])."""
    from refcases import CASES, check_mean_std
    case = CASES[0]
    pdf = lambda x: (-(x + 5) ** 2 / (2 * 3.0 ** 2), np.ones(1000))
    theta0s = kmc.make_theta0s(-4.0, 0.1, pdf, 100, hasblob=True, rng=5)
    samples = kmc.emcee(pdf, theta0s, niter=10 ** 4, hasblob=True, use_progress_meter=False, seed=8)
    assert [len(x) for x in samples] == [100, 100, 100, 100]
    assert samples[0].shape[1] == 10 ** 4 // 100 // 2
    thetas, ar, logd, blobs = kmc.squash_walkers(*samples, verbose=False)
    assert len(thetas) == len(logd) == 10 ** 4 // 2 and ar > 0.1
    check_mean_std(thetas, case)
    assert len(blobs) == 10 ** 4 // 2 and all(np.array_equal(b, np.ones(1000)) for b in blobs)

    def add(blobs, blob):
        blobs[0] += blob[0]

    samples = kmc.emcee(pdf, theta0s, niter=10 ** 4, hasblob=True, use_progress_meter=False, seed=9,
                        init_blobs=lambda blob0, nsamples: [0], reduce_blob=add)
    assert len(samples[3]) == 100 and all(b == [50] for b in samples[3])
    thetas, ar, logd, blobs = kmc.squash_walkers(*samples, verbose=False, merge_blobs=add)
    check_mean_std(thetas, case)
    assert blobs == [10 ** 4 // 2]


def test_vectorized_blobs_and_misuse(kmc):
    th = 0.3 * np.random.default_rng(4).standard_normal((16, 2))
    f = lambda X: (-0.5 * (X * X).sum(axis=1), [tuple(r) for r in X])
    pdf = kmc.HostLogPdf(f, vectorized=True, hasblob=True)
    thetas, acc, logd, blobs = kmc.emcee(pdf, th, niter=16 * 10, nburnin=0, hasblob=True, use_progress_meter=False, seed=2)
    for w in range(16):
        np.testing.assert_array_equal(np.array(blobs[w]), thetas[w])
    with pytest.raises(NotImplementedError, match="host callable"):
        kmc.emcee(kmc.GaussianIso(), th, niter=160, hasblob=True, use_progress_meter=False)
    with pytest.raises(ValueError, match="hasblob=True"):
        kmc.emcee(lambda x: 0.0, th, niter=160, init_blobs=lambda b, n: [], use_progress_meter=False)

    def boom(blobs, blob):
        raise KeyError("reduce")

    with pytest.raises(KeyError, match="reduce"):
        kmc.emcee(lambda x: (-0.5 * float(x @ x), 1), th, niter=160, nburnin=0, hasblob=True, reduce_blob=boom,
                  use_progress_meter=False)


def test_accept_outcomes_through_the_c_abi(kmc, oracle):
    """kmc_config.host_accepted: per half-step flags = the oracle's accept decisions (difference of its counters)."""
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    nw, nd, G = 20, 3, 12
    th = np.ascontiguousarray(0.3 * np.random.default_rng(1).standard_normal((nw, nd)))
    seen = []

    @_lib.HOST_LOGPDF_FN
    def cb(rows, nrows, ndim, out, user):
        X = np.ctypeslib.as_array(rows, shape=(nrows, ndim))
        for i in range(nrows):
            out[i] = oracle.logpdf(0, [0.0, 1.0], X[i])
        return 0

    @_lib.HOST_ACCEPTED_FN
    def acc_cb(flags, nrows, row0, generation, stored, user):
        seen.append((int(generation), int(row0), int(stored), np.ctypeslib.as_array(flags, shape=(nrows,)).copy()))
        return 0

    c = _lib.Config()
    c.dtype, c.density = _lib.F64, _lib.HOST_DENSITY
    c.nwalkers, c.ndim, c.ngenerations, c.nburnin, c.nthin, c.a_scale, c.seed = nw, nd, G, 4, 2, 2.0, 6
    c.host_logpdf = C.cast(cb, C.c_void_p)
    c.host_accepted = C.cast(acc_cb, C.c_void_p)
    nacc = np.zeros(nw, dtype=np.int64)
    out = _lib.Outputs()
    out.naccept = nacc.ctypes.data_as(C.POINTER(C.c_int64))
    _lib.check(L.kmc_emcee_run(C.byref(c), th.ctypes.data_as(C.POINTER(C.c_double)), C.byref(out)))
    assert [(g, r0) for g, r0, _, _ in seen] == [(g, r0) for g in range(G) for r0 in (0, nw // 2)]
    assert [st for _, _, st, _ in seen[::2]] == [int(g >= 4 and (g - 4 + 1) % 2 == 0) for g in range(G)]   # :268, n = g - nburnin + 1
    total = np.zeros(nw, dtype=np.int64)
    for g, r0, _, fl in seen:
        assert set(np.unique(fl)) <= {0, 1}
        if g >= 4:                                  # counters restart after burn-in (:285-288)
            total[r0:r0 + nw // 2] += fl
    np.testing.assert_array_equal(total, nacc)
    ref = oracle.emcee(oracle.make_config(0, [0.0, 1.0], nw, nd, G, 4, 2, 2.0, 6), th)
    np.testing.assert_array_equal(nacc, ref["naccept"])
    # the callback belongs to KMC_HOST_DENSITY
    c.density, c.host_logpdf = 0, None
    assert L.kmc_validate(C.byref(c)) == _lib.ERR_BAD_ARG


@pytest.mark.parametrize("zero_copy", ["0", "1"])
def test_small_batches_with_and_without_the_copies(kmc, oracle, monkeypatch, zero_copy, kmc_debug):
    """Up to 256 KiB of proposals the kernels address the page-locked host arrays directly (KMC_DEBUG=host-zerocopy); forced on for a batch
    the copies would take, and off for one they would not: the same chain as the device density either way."""
    kmc_debug.set("host-zerocopy", zero_copy)
    nw, nd, G = 600, 3, 60
    th = np.random.default_rng(3).standard_normal((nw, nd))
    host = kmc.HostLogPdf(lambda X: -0.5 * (X * X).sum(axis=1), vectorized=True)
    out = []
    for pdf in (host, kmc.GaussianIso(0.0, 1.0)):
        with kmc.Sampler(pdf, nw, nd, G, 10, 1, 2.0, 5, store_chain=True, use_graph=True) as s:
            s.set_positions(th)
            s.run(G)
            s.sync()
            out.append((s.chain(logp=False)[0], s.naccept()))
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][0], out[1][0])
