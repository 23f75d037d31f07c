"""GPU: the ends of the parameter ranges against the oracle -- ensembles of millions of walkers (32-bit index arithmetic),
stretch scales next to 1 and far above it, ensembles of exactly ndim + 2 walkers, runs of a single generation, thinning
beyond the run length."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(kmc, oracle, pdf, did, params, th, G, nburn, nthin, a, seed, **kw):
    nw, nd = th.shape
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, nthin, a, seed, nthreads=8), th, store_chain=False)
    assert ref["status"] == 0
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, a, seed, moments=True, **kw) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        np.testing.assert_array_equal(s.naccept(), ref["naccept"])
        np.testing.assert_array_equal(s.positions(), ref["final_pos"])
        tol = 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"]))
        assert np.all(np.abs(s.logp() - ref["final_logp"]) <= tol)
        msum, msq, n = s.moments()
        assert n == ref["nmoment"]
        np.testing.assert_allclose(msum, ref["sum"], rtol=1e-10, atol=1e-7)
        return s.describe()


@pytest.mark.parametrize("nw,nd", [(1 << 24, 1), (1 << 23, 2), (3_000_002, 5), (1 << 21, 32)])
def test_millions_of_walkers(kmc, oracle, nw, nd):
    th = np.random.default_rng(nd).standard_normal((nw, nd))
    _both(kmc, oracle, kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0], th, 3, 1, 1, 2.0, 77)


@pytest.mark.parametrize("a", [1.0000001, 1.01, 50.0, 1e4])
def test_extreme_stretch_scales(kmc, oracle, a):
    th = np.random.default_rng(3).standard_normal((512, 8))
    _both(kmc, oracle, kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0], th, 70, 10, 1, a, 5)
    _both(kmc, oracle, kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0], th[:100, :1].copy(), 70, 10, 1, a, 5)     # resident kernel


@pytest.mark.parametrize("nd", [1, 2, 7, 32, 62, 200, 1022, 1500])
def test_smallest_legal_ensemble(kmc, oracle, nd):
    """nwalkers = ndim + 2 (rounded up to even): the reference's lower bound (src/samplers.jl:205)."""
    nw = nd + 2 + (nd % 2)
    th = 0.5 * np.random.default_rng(nd).standard_normal((nw, nd))
    G = 40 if nd < 1000 else 6
    _both(kmc, oracle, kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0], th, G, G // 4, 1, 2.0, 9)
    with pytest.raises(kmc.KmcError, match="Use more walkers"):
        kmc.Sampler(kmc.GaussianIso(), nw - 2, nd, 10)


def test_degenerate_run_lengths(kmc, oracle):
    th = np.random.default_rng(4).standard_normal((256, 4))
    for G, nburn, nthin in [(1, 0, 1), (1, 1, 1), (5, 0, 7), (64, 63, 1), (65, 0, 64), (130, 1, 129)]:
        nw, nd = th.shape
        ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, nthin, 2.0, 3), th)
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, nthin, 2.0, 3, store_chain=True, store_logp=True, moments=True) as s:
            s.set_positions(th)
            s.run(G)
            s.sync()
            ch, lp = s.chain()
            assert ch.shape == ref["chain"].shape == ((G - nburn) // nthin, nw, nd), (G, nburn, nthin)
            np.testing.assert_array_equal(ch, ref["chain"])
            np.testing.assert_array_equal(s.naccept(), ref["naccept"])
            np.testing.assert_array_equal(s.positions(), ref["final_pos"])
            assert s.moments()[2] == ref["nmoment"]
