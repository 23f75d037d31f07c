"""GPU: KMC_STREAM_CHAIN -- the stored samples (reference src/samplers.jl:268-272: thetas[nc], logdensities[nc] grow by
push!) leave the device block by block while sampling goes on: a ring of three sample blocks in HBM, a second stream
copying completed blocks into the caller's (page-locked) host arrays.  The streamed chain must equal the oracle's chain
entry for entry, across several laps of the ring, whatever the launch mode, thinning, row padding or run splitting."""
import ctypes as C
import os

import numpy as np
import pytest

import kmcenv

pytestmark = pytest.mark.gpu


def _oracle_chain(oracle, did, params, th, G, nburn, nthin, seed):
    nw, nd = th.shape
    r = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, nthin, 2.0, seed, nthreads=8), th)
    assert r["status"] == 0
    return r


def _equal(chain, clogp, ref):
    np.testing.assert_array_equal(chain, ref["chain"])
    assert np.all(np.abs(clogp - ref["chain_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["chain_logp"])))


def test_streamed_chain_equals_oracle_over_three_ring_laps(kmc, oracle, monkeypatch, kmc_debug):
    """4096 x 32, nthin = 1, 1340 stored samples through a ring of 3 x 129 sample slots (a block holds a replay of 128 generations + 1): > 3 laps."""
    kmc_debug.set("chain-block", "1")            # smallest legal block: the samples of one graph replay (+1)
    nw, nd, G, nburn, seed = 4096, 32, 1400, 60, 5
    th = np.random.default_rng(1).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, seed, store_chain=True, store_logp=True, moments=True,
                     stream_chain=True) as s:
        assert "chain streamed to host memory in blocks of 129 samples" in s.describe()
        s.set_positions(th)
        s.run(G)
        s.sync()
        chain, clogp = s.chain()
        assert chain.shape == (G - nburn, nw, nd) and (G - nburn) > 3 * 3 * 129
        ref = _oracle_chain(oracle, oracle.GAUSSIAN_ISO, [0.0, 1.0], th, G, nburn, 1, seed)
        _equal(chain, clogp, ref)
        np.testing.assert_array_equal(s.positions(), ref["final_pos"])
        np.testing.assert_array_equal(s.naccept(), ref["naccept"])
        m = s.moments()
        np.testing.assert_allclose(m[0], ref["sum"], rtol=1e-11, atol=1e-8)


@pytest.mark.parametrize("case", ["thin3_odd_ndim", "pieces_with_syncs", "eager_launches", "logp_only", "chain_only", "rosen_draw_ring",
                                  "small_ensemble", "unregistered_destination", "two_walkers_per_thread"])
@pytest.mark.parametrize("by_walker", [False, True], ids=["sample-major", "by-walker"])
@pytest.mark.parametrize("resident", [True, False], ids=["resident-where-it-fits", "multi-launch"])
def test_streamed_chain_variants(kmc, oracle, monkeypatch, case, by_walker, resident, kmc_debug):
    """by-walker: KMC_STREAM_CHAIN | KMC_CHAIN_BY_WALKER -- completed blocks are written by a kernel of the copy stream
    into host arrays laid out [walker][nsamples][ndim] (the reference's thetas[w][k]).  Ensembles of up to 1024 walkers run in
    resident mode (launches cut to less than a block of the ring, ring positions carried by the kernel); KMC_DEBUG=no-resident keeps
    the same jobs on the multi-launch kernels."""
    kmc_debug.set("chain-block", "1")
    if not resident:
        kmcenv.no_resident(monkeypatch)
    pdf, did, params, nw, nd, G, nburn, nthin, scale = kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0], 1024, 8, 900, 37, 1, 1.0
    kw = dict(store_chain=True, store_logp=True)
    if case == "thin3_odd_ndim":
        nd, nthin, G = 7, 3, 1900                        # padded device rows -> strided copies; 23 samples per replay
    elif case == "eager_launches":
        kw["use_graph"] = False
    elif case == "logp_only":
        kw = dict(store_chain=False, store_logp=True)
    elif case == "chain_only":
        kw = dict(store_chain=True, store_logp=False)
    elif case == "rosen_draw_ring":
        pdf, did, params, nd, scale = kmc.Rosenbrock(), oracle.ROSENBROCK, [1.0, 100.0, 20.0], 64, 0.1
    elif case == "small_ensemble":
        nw, nd = 100, 2                                  # the reference's own size
    elif case == "two_walkers_per_thread":
        nw, nd = 1500, 3                                 # resident mode beyond 1024 walkers
    elif case == "unregistered_destination":
        kmc_debug.set("no-host-register")  # staged copies instead of DMA into page-locked arrays
    th = scale * np.random.default_rng(2).standard_normal((nw, nd))
    seed = 77
    # (by walker into arrays that cannot be page-locked -- a container's RLIMIT_MEMLOCK is enough: the transposed blocks come
    #  through the library's bounce buffers and are put in place by the host; never refused, kmc_emcee_run relies on it)
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, stream_chain=True, chain_by_walker=by_walker, **kw) as s:
        if resident and nd <= 32 and kw.get("use_graph", True):
            assert "resident mode" in s.describe() and "streamed" in s.describe(), s.describe()
        else:
            assert "resident" not in s.describe(), s.describe()
        s.set_positions(th)
        if case == "pieces_with_syncs":
            for n in (50, 1, 130, 64, 200, G - 445):      # syncs inside blocks: partial flush, then the whole block again
                s.run(n)
                s.sync()
                c, l = s.chain()
                assert len(c) == max(0, (s.generation - nburn) // nthin)
        else:
            s.run(G)
        s.sync()
        chain, clogp = s.chain()
        cw, lw = s.chain(by_walker=True)
        if chain is not None:
            np.testing.assert_array_equal(cw, chain.transpose(1, 0, 2))
        if clogp is not None:
            np.testing.assert_array_equal(lw, clogp.T)
    ref = _oracle_chain(oracle, did, params, th, G, nburn, nthin, seed)
    assert ref["nsamples"] > 6 * 129 // max(1, nthin) or nthin > 1
    if kw["store_chain"]:
        np.testing.assert_array_equal(chain, ref["chain"])
    else:
        assert chain is None
    if kw["store_logp"]:
        assert np.all(np.abs(clogp - ref["chain_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["chain_logp"])))
    else:
        assert clogp is None


def test_streamed_chain_restart_and_emcee_front_end(kmc, oracle, monkeypatch, kmc_debug):
    kmc_debug.set("chain-block", "1")
    nw, nd, G, nburn = 512, 4, 400, 100
    th = np.random.default_rng(3).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, 9, store_chain=True, store_logp=True, stream_chain=True) as s:
        for _ in range(2):                               # set_positions restarts the job: the ring bookkeeping too
            s.set_positions(th)
            s.run(G)
            s.sync()
        chain, clogp = s.chain()
    _equal(chain, clogp, _oracle_chain(oracle, oracle.GAUSSIAN_ISO, [0.0, 1.0], th, G, nburn, 1, 9))
    # the reference's front end: identical outputs with and without streaming
    a = kmc.emcee(kmc.GaussianIso(), th, niter=nw * G, seed=4, use_progress_meter=False, stream_chain=True)
    b = kmc.emcee(kmc.GaussianIso(), th, niter=nw * G, seed=4, use_progress_meter=False, stream_chain=False)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    # (log-densities to rounding: the streamed run uses the multi-launch kernels, which sum a row's terms lane-striped; the
    #  other runs resident, one walker per thread, summing in index order)
    assert np.all(np.abs(a[2] - b[2]) <= 1e-12 * np.maximum(1.0, np.abs(b[2])))


@pytest.mark.parametrize("by_walker", [False, True], ids=["sample-major", "by-walker"])
def test_one_shot_c_abi_streams_into_the_callers_buffers(kmc, oracle, monkeypatch, by_walker, kmc_debug):
    """kmc_emcee_run with KMC_STREAM_CHAIN: samples land in out->chain / out->chain_logp directly (with KMC_CHAIN_BY_WALKER
    laid out [walker][sample][dim])."""
    from kissmcmc_jl_amd import _lib
    kmc_debug.set("chain-block", "1")
    nw, nd, G, nburn, nthin, seed = 2048, 16, 460, 20, 2, 123
    th = np.ascontiguousarray(np.random.default_rng(5).standard_normal((nw, nd)))
    ns = (G - nburn) // nthin
    cfg = _lib.Config()
    cfg.dtype, cfg.density = _lib.F64, _lib.GAUSSIAN_ISO
    cfg.params[0], cfg.params[1] = 0.0, 1.0
    cfg.nwalkers, cfg.ndim, cfg.ngenerations, cfg.nburnin, cfg.nthin = nw, nd, G, nburn, nthin
    cfg.a_scale, cfg.seed, cfg.flags, cfg.device = 2.0, seed, _lib.STREAM_CHAIN | (_lib.CHAIN_BY_WALKER if by_walker else 0), 0
    dp = C.POINTER(C.c_double)
    chain = np.zeros((nw, ns, nd) if by_walker else (ns, nw, nd)); clogp = np.zeros((nw, ns) if by_walker else (ns, nw)); fpos = np.zeros((nw, nd))
    out = _lib.Outputs()
    out.chain, out.chain_logp, out.final_pos = chain.ctypes.data_as(dp), clogp.ctypes.data_as(dp), fpos.ctypes.data_as(dp)
    _lib.check(_lib.lib().kmc_emcee_run(C.byref(cfg), th.ctypes.data_as(dp), C.byref(out)))
    ref = _oracle_chain(oracle, oracle.GAUSSIAN_ISO, [0.0, 1.0], th, G, nburn, nthin, seed)
    assert out.nsamples == ns
    if by_walker:
        chain, clogp = chain.transpose(1, 0, 2), clogp.T
    _equal(chain, clogp, ref)
    np.testing.assert_array_equal(fpos, ref["final_pos"])


def test_stream_chain_argument_checks(kmc):
    with pytest.raises(kmc.KmcError, match="KMC_F64"):
        kmc.Sampler(kmc.GaussianIso(), 256, 4, 100, 10, store_chain=True, stream_chain=True, dtype="f32")
    with kmc.Sampler(kmc.GaussianIso(), 256, 4, 100, 10, store_chain=True) as s:      # not streaming
        from kissmcmc_jl_amd import _lib
        a = np.zeros((90, 256, 4))
        with pytest.raises(kmc.KmcError, match="KMC_STREAM_CHAIN"):
            _lib.check(s._L.kmc_sampler_set_chain_host(s._h, a.ctypes.data_as(C.POINTER(C.c_double)), None))


def test_streamed_chain_random_splits_and_thinnings(kmc, oracle, monkeypatch, kmc_debug):
    """Seeded random sweep of the ring bookkeeping: thinning, burn-in, block size, launch mode and the way a run is cut into
    run() calls (with and without syncs in between) -- every streamed chain equals the oracle's."""
    rng = np.random.default_rng(int(os.environ.get("KMC_FUZZ_BASE", 2026)))
    nw, nd = 512, 6
    th = rng.standard_normal((nw, nd))
    for trial in range(int(os.environ.get("KMC_FUZZ_TRIALS", 12))):
        nthin = int(rng.choice([1, 1, 2, 3, 7, 64, 65]))
        G = int(rng.integers(150, 700))
        nburn = int(rng.integers(0, G // 2))
        kmc_debug.set("chain-block", str(int(rng.choice([1, 3, 50, 1000]))))
        monkeypatch.setenv("KMC_LAUNCH", str(rng.choice(["graph", "updated", "eager"])))
        seed = int(rng.integers(1, 10 ** 6))
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, stream_chain=True,
                         chain_by_walker=bool(trial % 2)) as s:       # every other trial streams in the reference's order
            s.set_positions(th)
            left = G
            while left > 0:
                n = int(min(left, rng.choice([1, 5, 63, 64, 65, 128, 300])))
                s.run(n)
                left -= n
                if rng.random() < 0.4:
                    s.sync()
            s.sync()
            chain, clogp = s.chain()
        ref = _oracle_chain(oracle, oracle.GAUSSIAN_ISO, [0.0, 1.0], th, G, nburn, nthin, seed)
        assert chain.shape[0] == ref["nsamples"] == (G - nburn) // nthin, (trial, nthin, G, nburn)
        np.testing.assert_array_equal(chain, ref["chain"], err_msg=f"trial {trial}: nthin={nthin} G={G} nburn={nburn}")
        assert np.all(np.abs(clogp - ref["chain_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["chain_logp"])))


def test_chain_larger_than_the_device_is_an_error_not_a_crash(kmc):
    """A device-resident chain beyond HBM (C2 with 200 000 stored samples per walker: 336 GB): KMC_ERR_OOM from
    kmc_sampler_create, nothing leaks, and the next sampler works (kmc_emcee_run / emcee() switch to KMC_STREAM_CHAIN by
    themselves in this situation)."""
    from kissmcmc_jl_amd import _lib
    with pytest.raises(kmc.KmcError) as ei:
        kmc.Sampler(kmc.GaussianIso(), 65536, 32, 200000, 0, 1, store_chain=True)
    assert ei.value.status == _lib.ERR_OOM
    th = np.random.default_rng(0).standard_normal((256, 4))
    with kmc.Sampler(kmc.GaussianIso(), 256, 4, 20, 0, 1, store_chain=True) as s:
        s.set_positions(th)
        s.run(20)
        s.sync()
        assert s.chain(logp=False)[0].shape == (20, 256, 4)


def test_small_ensemble_streams_a_long_chain_from_resident_mode(kmc, oracle):
    """The reference's own size with the default block (4096 samples): launches of 1024 generations, several laps of the ring,
    thinning -- equal to the oracle, and about as fast as the unstreamed resident run."""
    nw, nd, G, nburn, nthin, seed = 100, 1, 40000, 1000, 2, 3
    th = 0.5 + 0.1 * np.abs(np.random.default_rng(8).standard_normal((nw, nd)))
    with kmc.Sampler(kmc.Exponential(1.0), nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, stream_chain=True) as s:
        assert "resident mode" in s.describe()
        s.set_positions(th)
        s.run(G)
        s.sync()
        chain, clogp = s.chain()
        assert s.launch_count <= 2 * (G // 1024 + 1)
        ms = s.last_run_ms()
    ref = _oracle_chain(oracle, oracle.EXPONENTIAL, [1.0], th, G, nburn, nthin, seed)
    assert chain.shape[0] == (G - nburn) // nthin > 3 * 4096
    _equal(chain, clogp, ref)
    assert ms < 2 * G * 1.0e-3, f"{ms:.1f} ms for {G} generations: not the resident kernel's speed"       # (< 1 us per half-step; 0.22 measured)
