"""GPU: device-side make_theta0s (kmc_sampler_init_ball) and checkpoint / resume (kmc_sampler_set_state)."""
import numpy as np
import pytest

import glob
import os

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
BALL_FIX = sorted(glob.glob(os.path.join(HERE, "golden", "init_ball", "*.npz")))
_DENS = {0: lambda k, p: k.GaussianIso(p[0], p[1]), 1: lambda k, p: k.Exponential(p[0]), 3: lambda k, p: k.LogNormal(p[0], p[1])}
# device log / sincos against glibc: <= a few ulp in each normal; positions = theta0 + normal * radius
BALL_RTOL, BALL_ATOL = 1e-11, 1e-13


def _ball_matches(pos, logp, ref_pos, ref_logp, scale=1.0):
    np.testing.assert_allclose(pos, ref_pos, rtol=BALL_RTOL, atol=BALL_ATOL * scale)
    np.testing.assert_allclose(logp, ref_logp, rtol=1e-10, atol=1e-10 * scale)


@pytest.mark.parametrize("path", BALL_FIX, ids=[os.path.basename(p)[:-4] for p in BALL_FIX])
def test_init_ball_equals_golden_fixture(kmc, path):
    """kmc_sampler_init_ball against the committed oracle output (tests/golden/init_ball): the same try is kept for every
    walker -- identical accept / retry / shrink decisions -- and its coordinates agree to rounding."""
    z = dict(np.load(path))
    nw, nd = int(z["nwalkers"]), int(z["ndim"])
    pdf = _DENS[int(z["density"])](kmc, z["params"])
    with kmc.Sampler(pdf, nw, nd, 10) as s:
        if int(z["nfail"]) > 0:
            with pytest.raises(RuntimeError, match=rf"Could not find suitable initial theta.*\({int(z['nfail'])} walkers\)"):
                s.init_ball(z["theta0"], z["radius"], seed=int(z["seed"]), ball_radius_halfing_steps=int(z["halving_steps"]), ntries=int(z["ntries"]))
            return
        s.init_ball(z["theta0"], z["radius"], seed=int(z["seed"]), ball_radius_halfing_steps=int(z["halving_steps"]), ntries=int(z["ntries"]))
        _ball_matches(s.positions(), s.logp(), z["pos"], z["logp"])


@pytest.mark.parametrize("case", ["gauss_65536x32", "expo_retry_4096x3", "expo_shrink_2048x8", "lognormal_odd_1000x7", "rosen_16384x64",
                                  "halving_40"])
def test_init_ball_equals_oracle(kmc, oracle, case):
    """The device-side make_theta0s (reference src/samplers.jl:311-349) against its CPU restatement kmco_init_ball on the
    same seed: positions within 1e-11 (a different try for any walker would be off by O(radius)), log-pdfs within 1e-10;
    including retries (pdf = -Inf inside the ball), the per-walker shrinking ball and halving_steps > 32 (formerly a
    shift by more than the word size)."""
    cfg = {
        "gauss_65536x32": (kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0], np.linspace(-1, 1, 32), np.linspace(0.05, 0.2, 32), 65536, 32, 7, 100, 42),
        "expo_retry_4096x3": (kmc.Exponential(), oracle.EXPONENTIAL, [1.0], 0.02, 0.1, 4096, 3, 7, 100, 7),
        "expo_shrink_2048x8": (kmc.Exponential(), oracle.EXPONENTIAL, [1.0], 0.01, 1.0, 2048, 8, 7, 3, 5),
        "lognormal_odd_1000x7": (kmc.LogNormal(), oracle.LOGNORMAL, [0.0, 1.0], 0.3, 0.5, 1000, 7, 7, 100, 9),
        "rosen_16384x64": (kmc.Rosenbrock(), oracle.ROSENBROCK, [1.0, 100.0, 20.0], 0.0, 0.1, 16384, 64, 7, 100, 1),
        "halving_40": (kmc.Exponential(), oracle.EXPONENTIAL, [1.0], 1e-160, 1.0, 512, 6, 40, 1, 13),   # admissible from k ~ 33 on
    }[case]
    pdf, did, params, th, rad, nw, nd, hs, nt, seed = cfg
    ref = oracle.init_ball(did, params, th, rad, nw, nd, seed=seed, halving_steps=hs, ntries=nt)
    assert ref["nfail"] == 0
    if "retry" in case or "shrink" in case or "halving" in case:
        assert (ref["attempts"] > 1).mean() > 0.3                     # the case does exercise retries
    if "shrink" in case:
        assert (ref["attempts"] > 2 * nt).any()                       # ... and the third ball size
    if "halving" in case:
        assert (ref["attempts"] > 32).mean() > 0.3                    # ball sizes beyond 1/2^31
    with kmc.Sampler(pdf, nw, nd, 10) as s:
        s.init_ball(th, rad, seed=seed, ball_radius_halfing_steps=hs, ntries=nt)
        _ball_matches(s.positions(), s.logp(), ref["pos"], ref["logp"], scale=1e-160 if "halving" in case else 1.0)
        s.run(4)                                                      # the sampler is ready to run
        s.sync()


@pytest.mark.parametrize("nw,nd,did", [(4096, 32, 0), (100, 2, 2)])     # multi-launch path; resident path (Rosenbrock)
def test_checkpoint_resume_equals_the_oracle_run(kmc, oracle, nw, nd, did):
    """state() at generation 130 -> restore() into a NEW sampler -> the rest of the run: final positions and acceptance
    counters must be those of the ORACLE's uninterrupted run (not merely of another HIP run)."""
    pdf = kmc.GaussianIso() if did == 0 else kmc.Rosenbrock()
    params = [0.0, 1.0] if did == 0 else [1.0, 100.0, 20.0]
    th = (1.0 if did == 0 else 0.1) * np.random.default_rng(4).standard_normal((nw, nd))
    G, nburn, seed = 200, 50, 21
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, 1, 2.0, seed), th, store_chain=False)
    assert ref["status"] == 0
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed, moments=True) as b:
        b.set_positions(th)
        b.run(130)
        b.sync()
        ck = b.state()
        m1 = b.moments()
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed, moments=True) as c:
        c.restore(ck)
        c.run(G - 130)
        c.sync()
        np.testing.assert_array_equal(c.positions(), ref["final_pos"])
        np.testing.assert_array_equal(c.naccept(), ref["naccept"])
        assert np.all(np.abs(c.logp() - ref["final_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"])))
        m2 = c.moments()
        assert m1[2] + m2[2] == ref["nmoment"]
        np.testing.assert_allclose(m1[0] + m2[0], ref["sum"], rtol=1e-11, atol=1e-8)
        np.testing.assert_allclose(m1[1] + m2[1], ref["sumsq"], rtol=1e-11, atol=1e-8)


def test_init_ball_statistics_and_determinism(kmc):
    """reference src/samplers.jl:311-349: nwalkers draws from N(theta0, diag(r^2)), all with pdf > -Inf."""
    nw, nd = 65536, 32
    theta0 = np.linspace(-1.0, 1.0, nd)
    radius = np.linspace(0.05, 0.2, nd)
    out = []
    for _ in range(2):
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, 10, moments=False) as s:
            s.init_ball(theta0, radius, seed=42)
            out.append((s.positions(), s.logp()))
    pos, logp = out[0]
    np.testing.assert_array_equal(pos, out[1][0])                     # seeded: reproducible
    assert np.all(np.isfinite(logp))
    np.testing.assert_allclose(logp, -0.5 * (pos ** 2).sum(axis=1), rtol=1e-12)
    np.testing.assert_allclose(pos.mean(axis=0), theta0, atol=5 * radius.max() / np.sqrt(nw))
    np.testing.assert_allclose(pos.std(axis=0), radius, rtol=0.02)
    z = (pos - theta0) / radius                                       # independent across walkers and dimensions
    c = np.corrcoef(z[:, :8].T)
    assert np.abs(c - np.eye(8)).max() < 0.02
    from scipy import stats
    assert stats.kstest(z[:, 5], "norm").pvalue > 1e-3


def test_init_ball_retries_and_shrinks_like_the_reference(kmc):
    """Exponential support x >= 0, ball centred at 0.02 with radius 0.1: ~42 % of first tries fail."""
    with kmc.Sampler(kmc.Exponential(), 4096, 3, 10) as s:
        s.init_ball(0.02, 0.1, seed=7)
        pos = s.positions()
        assert np.all(pos >= 0.0) and np.all(np.isfinite(s.logp()))
        assert pos.max() > 0.2                                        # not collapsed onto theta0
        s.run(10)
        s.sync()
    with kmc.Sampler(kmc.Exponential(), 64, 2, 10) as s:
        with pytest.raises(RuntimeError, match="Could not find suitable initial theta"):
            s.init_ball(-50.0, 0.1, seed=1)                           # src/samplers.jl:345 (intended)


def test_init_ball_with_user_density_and_small_ensemble(kmc):
    pdf = kmc.ExprDensity("x < 0.0 ? -INFINITY : -x")
    with kmc.Sampler(pdf, 100, 1, 1000, 500, moments=True) as s:
        s.init_ball(0.5, 0.1, seed=3)
        assert np.all(s.positions() >= 0)
        s.run(1000)
        s.sync()
        m = s.moments()
        assert abs(m[0][0] / m[2] - 1.0) < 0.1


@pytest.mark.parametrize("nw,nd", [(512, 32), (100, 2)])        # multi-launch path and resident path
def test_checkpoint_resume_is_bit_identical(kmc, nw, nd):
    th = np.random.default_rng(0).standard_normal((nw, nd))
    G, nburn = 200, 50
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, 11, moments=True) as a:
        a.set_positions(th)
        a.run(G)
        a.sync()
        ref = dict(pos=a.positions(), logp=a.logp(), nacc=a.naccept(), mom=a.moments())
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, 11, moments=True) as b:
        b.set_positions(th)
        b.run(130)
        b.sync()
        ck = b.state()
        m1 = b.moments()
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, 11, moments=True) as c:
        c.restore(ck)
        assert c.generation == 130
        c.run(G - 130)
        c.sync()
        np.testing.assert_array_equal(c.positions(), ref["pos"])
        np.testing.assert_array_equal(c.logp(), ref["logp"])
        np.testing.assert_array_equal(c.naccept(), ref["nacc"])
        m2 = c.moments()
        assert m1[2] + m2[2] == ref["mom"][2]                         # moments restart at the checkpoint
        np.testing.assert_allclose(m1[0] + m2[0], ref["mom"][0], rtol=1e-11, atol=1e-8)
        np.testing.assert_allclose(m1[1] + m2[1], ref["mom"][1], rtol=1e-11, atol=1e-8)


def test_samplers_give_their_device_memory_back(kmc):
    """Every kind of sampler, created, run and destroyed a few times over: the device's free memory afterwards is what it was
    before (kmc_device_free_bytes) -- the destroy path knows every buffer, event and stream the create path made, including the
    ones behind blobs, streamed chains, the host route's pieces, P2P shards and the resident kernels."""
    import ctypes as C
    from kissmcmc_jl_amd import _lib

    def free_bytes():
        f, t = C.c_uint64(0), C.c_uint64(0)
        _lib.check(_lib.lib().kmc_device_free_bytes(0, C.byref(f), C.byref(t)))
        return f.value

    nd = 8
    th = lambda nw: np.random.default_rng(1).standard_normal((nw, nd))
    blob = kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; blob[0] = x[0]; blob[1] = s; return -0.5 * s;", nblob=2)
    host = kmc.HostLogPdf(lambda X: -0.5 * (X * X).sum(axis=1), vectorized=True)
    kinds = [
        (kmc.GaussianIso(), 4096, dict(store_chain=True, store_logp=True, moments=True)),
        (kmc.GaussianIso(), 100, dict(store_chain=True, moments=True)),                              # resident, one walker per thread
        (kmc.GaussianIso(), 4096, dict(store_chain=True, store_logp=True, stream_chain=True, chain_by_walker=True)),
        (kmc.GaussianIso(), 4096, dict(moments=True, island_gens=8, island_size=64)),
        (kmc.GaussianIso(), 4096, dict(store_chain=True, dtype="f32")),
        (blob, 4096, dict(store_chain=True, store_blobs=True)),
        (blob, 128, dict(store_chain=True, store_blobs=True)),                                       # resident with blobs
        (kmc.ExprDensity("-0.5 * x * x"), 4096, dict(moments=True)),
        (host, 16384, dict(store_chain=True)),                                                       # proposals in pieces: events
        (kmc.GaussianIso(), 4096, dict(moments=True, deal_rank=0, deal_count=2)),
    ]

    def cycle():
        for pdf, nw, kw in kinds:
            with kmc.Sampler(pdf, nw, nd, 80, 10, 1, 2.0, 3, **kw) as s:
                s.set_positions(th(nw))
                s.run(80 if not kw.get("island_gens") else 64)
                s.sync()
        shards = [kmc.Sampler(kmc.GaussianIso(), 2048, nd, 64, 0, 1, 2.0, 3, p2p=True, shard_rank=r, shard_count=2) for r in range(2)]
        kmc.Sampler.p2p_connect_local(shards)
        for sh in shards:
            sh.close()
        from kissmcmc_jl_amd.metropolis import run_chains
        run_chains(blob, kmc.GaussianStep(0.5), th(512), 60, 10, 1, 4, store_blobs=True)

    cycle()                                         # first use: code objects, graph pools, bounce buffers stay with the process
    base = free_bytes()
    for _ in range(3):
        cycle()
    drift = base - free_bytes()
    assert abs(drift) <= 32 << 20, f"device memory drifted by {drift / 2 ** 20:.1f} MiB over three cycles"


def test_small_buffers_are_recycled_and_can_be_released(kmc):
    """The allocation cache (kmc_host.hpp: cache_alloc): samplers of one shape reuse each other's blocks -- what the device reports
    free does not move between the second and the tenth sampler --, results do not depend on what a recycled block held, and
    kmc_device_cache_release hands everything back."""
    import ctypes as C
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()

    def raw_free():                                   # hipMemGetInfo itself, without the cache's blocks counted in
        import torch
        return torch.cuda.mem_get_info(0)[0]

    def one(seed):
        with kmc.Sampler(kmc.GaussianIso(), 100, 3, 200, 50, 1, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
            s.set_positions(np.random.default_rng(4).standard_normal((100, 3)))
            s.run(200)
            s.sync()
            return s.positions(), s.chain()[0], s.moments()

    first = one(5)
    one(6)
    f0 = raw_free()
    for k in range(8):
        one(7 + k)
    assert abs(raw_free() - f0) <= 4 << 20, "same-shaped samplers should reuse cached blocks"
    again = one(5)                                    # recycled blocks, dirty with other runs' data: same results
    for a, b in zip(first, again):
        if isinstance(a, tuple):
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)
        else:
            np.testing.assert_array_equal(a, b)
    L.kmc_device_cache_release()
    assert raw_free() >= f0, "released blocks go back to the device"
    np.testing.assert_array_equal(one(5)[1], first[1])
