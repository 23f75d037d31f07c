"""Host logic (no GPU): the reference's bookkeeping, make_theta0s, squash_walkers and the
argument checks of emcee(), compared with the oracle's line-by-line restatements."""
import numpy as np
import pytest

from oracle import host as ohost


def test_emcee_counts_follow_reference_divisions(kmc):
    # README call: niter=10^5, 100 walkers -> 1000 generations, 500 burn-in, 500 samples
    assert kmc.emcee_counts(10 ** 5, 100) == (1000, 500, 500)
    for niter, nw, nb, nthin in [(10 ** 4, 100, None, 1), (10 ** 5, 100, 0, 3), (12345, 98, 777, 2), (50, 100, None, 1)]:
        assert kmc.emcee_counts(niter, nw, nb, nthin) == ohost.emcee_counts(niter, nw, nb, nthin)


@pytest.mark.parametrize("theta0,ball", [(0.5, 0.1), ([0.0, 0.0], 0.1), ([0.4, 0.3, -1.0], [0.1, 0.2, 0.3])])
def test_make_theta0s_equals_reference_restatement(kmc, theta0, ball):
    pdf = kmc.Exponential() if np.ndim(theta0) == 0 else kmc.GaussianIso()
    got = kmc.make_theta0s(theta0, ball, pdf, 100, rng=np.random.default_rng(7))
    ref = ohost.make_theta0s(theta0, ball, pdf, 100, np.random.default_rng(7))
    assert got.shape == ((100,) if np.ndim(theta0) == 0 else (100, len(theta0)))
    np.testing.assert_array_equal(got, np.array(ref))


def test_make_theta0s_retries_like_the_reference(kmc):
    """Half of the first tries land outside the support -> the sequential retry path runs."""
    pdf = kmc.Exponential()
    got = kmc.make_theta0s(0.0, 0.1, pdf, 60, rng=np.random.default_rng(3))
    ref = ohost.make_theta0s(0.0, 0.1, pdf, 60, np.random.default_rng(3))
    np.testing.assert_array_equal(got, np.array(ref))
    assert np.all(got >= 0) and len(got) == 60
    got2 = kmc.make_theta0s([0.05, 0.05], 0.1, pdf, 40, rng=np.random.default_rng(4))
    ref2 = ohost.make_theta0s([0.05, 0.05], 0.1, pdf, 40, np.random.default_rng(4))
    np.testing.assert_array_equal(got2, np.array(ref2))


def test_make_theta0s_errors(kmc):
    with pytest.raises(AssertionError):
        kmc.make_theta0s([0.0, 0.0], [0.1, 0.1, 0.1], kmc.GaussianIso(), 10)      # samplers.jl:319
    with pytest.raises(RuntimeError, match="Could not find suitable initial theta"):
        kmc.make_theta0s(-50.0, 0.1, kmc.Exponential(), 4, rng=1)                 # samplers.jl:345 (intended)
    with pytest.raises(NotImplementedError):
        kmc.make_theta0s(0.5, 0.1, kmc.Exponential(), 4, hasblob=True)


def _fake_run(nw, ns, nd, seed=0):
    rng = np.random.default_rng(seed)
    thetas = rng.standard_normal((nw, ns, nd)) if nd else rng.standard_normal((nw, ns))
    acc = rng.uniform(0.2, 0.3, nw)
    acc[3] = 0.01
    logd = rng.standard_normal((nw, ns))
    return thetas, acc, logd


@pytest.mark.parametrize("nd", [0, 2])
@pytest.mark.parametrize("order", [False, True])
@pytest.mark.parametrize("drop", [False, True])
def test_squash_walkers_equals_reference_restatement(kmc, nd, order, drop):
    thetas, acc, logd = _fake_run(10, 7, nd)
    t, a, l, b = kmc.squash_walkers(thetas, acc, logd, drop_low_accept_ratio=drop, order=order, verbose=False)
    as_lists = [[tuple(np.atleast_1d(x)) for x in w] for w in thetas]
    rt, ra, rl, rb = ohost.squash_walkers(as_lists, list(acc), [list(w) for w in logd],
                                          drop_low_accept_ratio=drop, order=order)
    np.testing.assert_array_equal(np.asarray(t).reshape(len(rt), -1), np.array(rt).reshape(len(rt), -1))
    np.testing.assert_array_equal(l, np.array(rl))
    assert a == pytest.approx(ra, rel=1e-15) and b is None and rb is None
    assert len(t) == (9 if drop else 10) * 7


def test_squash_walkers_positional_like_the_reference(kmc):
    """squash_walkers(samples...) -- test/emcee.jl:36 splats emcee's 4-tuple."""
    thetas, acc, logd = _fake_run(6, 5, 0)
    t, a, l, b = kmc.squash_walkers(*(thetas, acc, logd, None), verbose=False)
    assert t.shape == (30,) and l.shape == (30,) and b is None
    t2, a2, l2, b2 = kmc.squash_walkers(thetas, acc)        # README.md:27
    assert l2 is None and a2 == pytest.approx(acc.mean())
    np.testing.assert_array_equal(t2[:5], thetas[0])        # walker-major (samplers.jl:398-399)


def test_emcee_argument_checks_match_reference_asserts(kmc):
    g = kmc.GaussianIso()
    th = np.zeros((10, 2))
    with pytest.raises(AssertionError):
        kmc.emcee(g, th, a_scale=1.0, use_progress_meter=False)                         # samplers.jl:200
    with pytest.raises(AssertionError, match="Use an even number of walkers."):
        kmc.emcee(g, np.zeros((11, 2)), use_progress_meter=False)                        # :202
    with pytest.raises(AssertionError, match="Use more walkers: at least DOF\\+2"):
        kmc.emcee(g, np.zeros((2, 2)), use_progress_meter=False)                         # :205
    with pytest.raises(TypeError, match="callable"):
        kmc.emcee("not a density", th, use_progress_meter=False)
    from kissmcmc_jl_amd import _lib
    if _lib.lib().kmc_device_count() == 0:
        # a closure is evaluated on the host, but the sampler itself has no CPU fallback
        with pytest.raises(kmc.KmcError) as e:
            kmc.emcee(lambda x: -np.sum(x ** 2), th, use_progress_meter=False)
        assert e.value.status == _lib.ERR_NO_DEVICE
    with pytest.raises(NotImplementedError):
        kmc.emcee(g, th, hasblob=True)


def test_density_host_formulas_match_oracle(kmc, oracle):
    rng = np.random.default_rng(0)
    cases = [(kmc.GaussianIso(-5, 3), oracle.GAUSSIAN_ISO), (kmc.Exponential(2.0), oracle.EXPONENTIAL),
             (kmc.Rosenbrock(), oracle.ROSENBROCK), (kmc.LogNormal(0.1, 0.7), oracle.LOGNORMAL),
             (kmc.MvNormal2([0.5, -0.25], [[0.47, 1.8], [1.8, 7.0]]), oracle.MVNORMAL2)]
    for pdf, did in cases:
        nd = 2 if did == oracle.MVNORMAL2 else 5
        for _ in range(20):
            x = rng.standard_normal(nd)
            if did in (oracle.EXPONENTIAL, oracle.LOGNORMAL) and rng.random() < 0.7:
                x = np.abs(x) + 0.01
            a, b = pdf(x), oracle.logpdf(did, pdf.params(), x)
            assert (a == b) or abs(a - b) <= 1e-12 * max(1.0, abs(b)), (pdf, x, a, b)
        X = rng.standard_normal((50, nd))
        np.testing.assert_array_equal(pdf.finite_rows(X), np.array([pdf(r) > -np.inf for r in X]))


def test_local_to_global_reassembles_p2p_shards(kmc):
    """P2P local order (first-half slice, then second-half slice per rank) -> global walker order."""
    from kissmcmc_jl_amd.distributed import local_to_global, shard_slice
    nw, world = 48, 4
    glob = np.arange(nw * 3, dtype=np.float64).reshape(nw, 3)
    h = nw // 2
    parts = []
    for r in range(world):
        b, n = shard_slice(nw, r, world)
        parts.append(np.concatenate([glob[b:b + n], glob[h + b:h + b + n]]))
    np.testing.assert_array_equal(local_to_global(parts, nw, world), glob)
    np.testing.assert_array_equal(local_to_global([glob], nw, 1), glob)


def test_expr_density_host_interface(kmc):
    """ExprDensity objects carry the id/params the C ABI needs (compilation itself: test_c_abi.py)."""
    from kissmcmc_jl_amd import _lib
    d = kmc.ExprDensity("-0.5*p[0]*x*x", params=[2.0])
    assert d.density_id == _lib.USER_DENSITY and d.params() == [2.0] and d.user_handle is not None
    assert "ExprDensity" in repr(d)


def test_squash_walkers_blobs_and_make_theta0s_hasblob(kmc):
    """src/samplers.jl:408-421 (blobs through squash_walkers) and :333-337 (make_theta0s with a (p, blob) pdf)."""
    rng = np.random.default_rng(0)
    thetas = rng.standard_normal((6, 4))
    acc = np.array([0.3, 0.31, 0.29, 0.3, 0.01, 0.3])
    blobs = [[(w, k) for k in range(4)] for w in range(6)]
    t, a, l, b = kmc.squash_walkers(thetas, acc, None, blobs, verbose=False)
    assert b == [(w, k) for w in range(6) for k in range(4)] and blobs[0] == [(0, k) for k in range(4)]   # deepcopy :411
    t, a, l, b = kmc.squash_walkers(thetas, acc, None, blobs, verbose=False, order=True, drop_low_accept_ratio=True, drop_fact=1)
    assert b == [(w, k) for k in range(4) for w in (0, 1, 2, 3, 5)]
    np.testing.assert_array_equal(t, thetas[[0, 1, 2, 3, 5]].T.reshape(-1))
    sums = [[float(w)] for w in range(6)]

    def add(b1, b2):
        b1[0] += b2[0]

    assert kmc.squash_walkers(thetas, acc, None, sums, verbose=False, merge_blobs=add)[3] == [15.0] and sums[0] == [0.0]
    th = kmc.make_theta0s(0.5, 0.1, lambda x: (-x if x >= 0 else -np.inf, "blob"), 10, hasblob=True, rng=3)
    assert th.shape == (10,) and np.all(th >= 0)
