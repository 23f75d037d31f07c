"""GPU: the chain in the reference's order -- thetas[w][k], logdensities[w][k] (src/samplers.jl:219-221, :268-272) --
from kmc_sampler_get_chain_by_walker (transposed on the device) equals the sample-major chain reordered on the host,
for every kernel family that stores a chain, partial runs, float rows, odd ndim, and pieces smaller than the ensemble."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nw,nd,G,nburn,nthin,kw", [
    (100, 1, 300, 100, 1, {}),                      # README shape: resident kernel, scalar walkers
    (256, 7, 90, 20, 3, {}),                        # odd ndim (padded rows on the device)
    (2048, 32, 150, 22, 4, {}),                     # multi-launch kernels, graph chunks + eager tail
    (2048, 32, 150, 22, 4, dict(dtype="f32")),      # float rows on the device, double on the host
    (600, 200, 40, 10, 2, {}),
    (4096, 8, 70, 0, 1, dict(use_graph=False)),
])
def test_by_walker_equals_reordered_sample_major(kmc, nw, nd, G, nburn, nthin, kw):
    th = np.random.default_rng(nd).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, nthin, 2.0, 17, store_chain=True, store_logp=True, **kw) as s:
        s.set_positions(th)
        for upto in (G // 3, G):                    # a partial run first: k = samples stored so far
            s.run(upto - s.generation)
            s.sync()
            a, la = s.chain(by_walker=True)
            b, lb = s.chain()
            k = b.shape[0]
            assert a.shape == (nw, k, nd) and la.shape == (nw, k)
            np.testing.assert_array_equal(a, b.transpose(1, 0, 2))
            np.testing.assert_array_equal(la, lb.T)
        assert k == (G - nburn) // nthin
        only, none = s.chain(logp=False, by_walker=True)
        assert none is None
        np.testing.assert_array_equal(only, a)


def test_by_walker_in_small_pieces(kmc, monkeypatch, kmc_debug):
    """The transposition runs in pieces of walkers that fit a scratch buffer; KMC_DEBUG=by-walker-piece-mb shrinks it so that a
    small chain takes many pieces (one walker per piece at the end of the range)."""
    nw, nd, G = 512, 16, 64
    th = np.random.default_rng(2).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 0, 1, 2.0, 3, store_chain=True, store_logp=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        b, lb = s.chain()
        for mb in ("0.05", "0.008", "0.001"):       # 6 walkers, 1 walker, less than one walker (-> one)
            kmc_debug.set("by-walker-piece-mb", mb)
            a, la = s.chain(by_walker=True)
            np.testing.assert_array_equal(a, b.transpose(1, 0, 2))
            np.testing.assert_array_equal(la, lb.T)


def test_emcee_returns_the_reference_layout(kmc, oracle):
    """kmc.emcee: thetas[w][k] straight from the device transposition = the oracle's chain, reordered."""
    nw, nd, niter = 128, 4, 128 * 60
    th = np.random.default_rng(5).standard_normal((nw, nd))
    thetas, acc, logd, _ = kmc.emcee(kmc.GaussianIso(), th, niter=niter, nthin=2, use_progress_meter=False, seed=21)
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, 60, 30, 2, 2.0, 21), th)
    np.testing.assert_array_equal(thetas, ref["chain"].transpose(1, 0, 2))
    np.testing.assert_array_equal(acc, ref["accept_ratio"])
    assert logd.shape == (nw, 15)
    sq = kmc.squash_walkers(thetas, acc, logd, verbose=False)
    np.testing.assert_array_equal(sq[0], thetas.reshape(-1, nd))          # walker-major concatenation (:398-399)
    assert np.shares_memory(sq[0], thetas)                                # ... is a view of what the device delivered


@pytest.mark.parametrize("host", [False, True], ids=["device-chains", "host-closures"])
def test_metropolis_chains_by_chain(kmc, host):
    """kmc_metropolis_run with KMC_CHAIN_BY_WALKER (what metropolis() / metropolis_chains() return: thetas[chain][sample])
    equals the sample-major chain, reordered; both routes (chains in one kernel; closures over one host round trip)."""
    from kissmcmc_jl_amd.metropolis import run_chains
    th = np.random.default_rng(4).standard_normal((96 if host else 3000, 3))
    pdf = (lambda x: -0.5 * float(np.dot(x, x))) if host else kmc.GaussianIso()
    niter = 40 if host else 200
    a = run_chains(pdf, kmc.GaussianStep(0.7), th, niter, 10, 3, 5, by_chain=True)
    b = run_chains(pdf, kmc.GaussianStep(0.7), th, niter, 10, 3, 5)
    assert a["chain"].shape == (th.shape[0], (niter - 10) // 3, 3)
    np.testing.assert_array_equal(a["chain"], b["chain"].transpose(1, 0, 2))
    np.testing.assert_array_equal(a["chain_logp"], b["chain_logp"].T)
    np.testing.assert_array_equal(a["naccept"], b["naccept"])
