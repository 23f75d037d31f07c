"""GPU: BASELINE.json's full-size configurations.

Where the oracle finishes in seconds (a few dozen generations at full size) the comparison is
direct and bit-exact; long runs are checked through size-independent properties: analytic
posterior moments and acceptance rate (north-star tolerance: 1 %), affine invariance of the
stretch move, invariance under launch geometry, graph replay and walker sharding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOGP_RTOL = 1e-12


def _run(kmc, pdf, th, G, nburn, seed, **kw):
    nw, nd = th.shape
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed, moments=True, **kw) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        m = s.moments()
        return dict(pos=s.positions(), logp=s.logp(), nacc=s.naccept(), sum=m[0], sumsq=m[1], n=m[2])


def _check_vs_oracle(oracle, did, params, th, G, nburn, seed, got):
    nw, nd = th.shape
    cfg = oracle.make_config(did, params, nw, nd, G, nburn, 1, 2.0, seed, nthreads=8)
    ref = oracle.emcee(cfg, th, store_chain=False)
    assert ref["status"] == 0
    np.testing.assert_array_equal(got["nacc"], ref["naccept"])
    np.testing.assert_array_equal(got["pos"], ref["final_pos"])
    assert np.all(np.abs(got["logp"] - ref["final_logp"]) <= LOGP_RTOL * np.maximum(1.0, np.abs(ref["final_logp"])))
    assert got["n"] == ref["nmoment"]
    np.testing.assert_allclose(got["sum"], ref["sum"], rtol=1e-10, atol=1e-7)
    np.testing.assert_allclose(got["sumsq"], ref["sumsq"], rtol=1e-10, atol=1e-7)


def test_c2_full_size_matches_oracle(kmc, oracle):
    """65 536 x 32 Gaussian, 150 generations (two graph replays + eager tail): bit-exact."""
    th = np.random.default_rng(2).standard_normal((65536, 32))
    got = _run(kmc, kmc.GaussianIso(), th, 150, 50, 12345)
    _check_vs_oracle(oracle, oracle.GAUSSIAN_ISO, [0.0, 1.0], th, 150, 50, 12345, got)


def test_c3_full_size_matches_oracle(kmc, oracle):
    """16 384 x 64 chained Rosenbrock/20 (divergent / low-accept stress)."""
    th = 0.1 * np.random.default_rng(3).standard_normal((16384, 64))
    got = _run(kmc, kmc.Rosenbrock(), th, 100, 30, 777)
    _check_vs_oracle(oracle, oracle.ROSENBROCK, [1.0, 100.0, 20.0], th, 100, 30, 777, got)
    assert 0.01 < got["nacc"].mean() / 70 < 0.6


def test_c5_full_size_matches_oracle(kmc, oracle):
    """8 192 x 1 024 Gaussian (HBM-bound regime, one walker per wave)."""
    th = np.random.default_rng(5).standard_normal((8192, 1024))
    got = _run(kmc, kmc.GaussianIso(), th, 12, 4, 4242)
    _check_vs_oracle(oracle, oracle.GAUSSIAN_ISO, [0.0, 1.0], th, 12, 4, 4242, got)


@pytest.mark.parametrize("nw,nd,G", [(2097152, 32, 70), (524288, 128, 66)])
def test_hbm_resident_shapes_match_oracle(kmc, oracle, nw, nd, G):
    """The two shapes bench.py times as `other_configs.HBM_*` -- state 512 MiB, beyond the 256 MiB Infinity Cache, the only lines of
    the bench that are served from HBM -- at full size: one graph replay + an eager tail, bit-exact against the oracle (moments on)."""
    th = np.random.default_rng(nd).standard_normal((nw, nd))
    got = _run(kmc, kmc.GaussianIso(), th, G, 3, 2024)
    _check_vs_oracle(oracle, oracle.GAUSSIAN_ISO, [0.0, 1.0], th, G, 3, 2024, got)


def test_c2_written_as_a_function_body_is_the_menu_run(kmc):
    """The C2 ensemble with the Gaussian written as a C function body (a sum over elements: recognised, lane-striped, built by the offline
    compiler at this size): the same arithmetic per element in the same lane order as the menu density, so positions, counters AND
    log-pdfs equal the menu run bit for bit -- the caller's closure pdf(theta) of src/samplers.jl:257 costs nothing in fidelity."""
    th = np.random.default_rng(2).standard_normal((65536, 32))
    body = kmc.CDensity("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;")
    assert body.separable
    got = _run(kmc, body, th, 200, 50, 12345)
    ref = _run(kmc, kmc.GaussianIso(), th, 200, 50, 12345)
    for k in ("pos", "nacc", "logp"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    assert got["n"] == ref["n"]
    np.testing.assert_array_equal(got["sum"], ref["sum"])


def test_c1_readme_shape_matches_oracle(kmc, oracle):
    """100 walkers x 1-D exponential, niter = 10^5 -> 1000 generations, 500 burn-in."""
    th = 0.5 + 0.1 * np.abs(np.random.default_rng(1).standard_normal((100, 1)))
    got = _run(kmc, kmc.Exponential(), th, 1000, 500, 9)
    _check_vs_oracle(oracle, oracle.EXPONENTIAL, [1.0], th, 1000, 500, 9, got)


def test_c2_long_run_moments_and_acceptance(kmc):
    """10^4-generation C2 job started at stationarity: posterior mean/variance within 1 % of the
    analytic values (the CPU reference's limit), acceptance 0.234 within 1 %."""
    th = np.random.default_rng(7).standard_normal((65536, 32))
    G, nburn = 10000, 5000
    got = _run(kmc, kmc.GaussianIso(), th, G, nburn, 2024)
    assert got["n"] == 65536 * (G - nburn)
    mean = got["sum"] / got["n"]
    var = got["sumsq"] / got["n"] - mean ** 2
    assert np.all(np.abs(mean) < 0.01), np.abs(mean).max()          # 1 % of sigma
    assert np.all(np.abs(var - 1.0) < 0.01), (var.min(), var.max())
    acc = got["nacc"] / (G - nburn)
    assert abs(acc.mean() - 0.234) < 0.00234 * 2
    assert 0.005 < acc.std() < 0.02                                  # per-walker spread (binomial + mixing)
    pos = got["pos"]                                                  # final ensemble is itself a posterior draw
    assert abs(pos.mean()) < 0.01 and abs(pos.var() - 1.0) < 0.01


def test_affine_invariance_at_full_size(kmc):
    """Goodman & Weare: the stretch move commutes with affine maps.  Sampling N(mu, sigma^2) from
    sigma*x0+mu must make the same accept decisions and give sigma*x+mu (to rounding)."""
    th = np.random.default_rng(11).standard_normal((65536, 32))
    a = _run(kmc, kmc.GaussianIso(0.0, 1.0), th, 40, 0, 5)
    b = _run(kmc, kmc.GaussianIso(3.0, 2.0), 2.0 * th + 3.0, 40, 0, 5)
    np.testing.assert_array_equal(a["nacc"], b["nacc"])
    np.testing.assert_allclose(b["pos"], 2.0 * a["pos"] + 3.0, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b["logp"], a["logp"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("plan", ["16,1,4", "8,2,1", "4,4,2"])
def test_geometry_and_graph_invariance_at_full_size(kmc, plan, monkeypatch):
    th = np.random.default_rng(13).standard_normal((65536, 32))
    base = _run(kmc, kmc.GaussianIso(), th, 130, 10, 6)
    monkeypatch.setenv("KMC_PLAN", plan)
    other = _run(kmc, kmc.GaussianIso(), th, 130, 10, 6, use_graph=False)
    np.testing.assert_array_equal(base["pos"], other["pos"])
    np.testing.assert_array_equal(base["nacc"], other["nacc"])
    np.testing.assert_allclose(base["sum"], other["sum"], rtol=1e-11, atol=1e-8)


def test_c4_shape_eight_logical_shards_equal_unsharded(kmc):
    """524 288 x 32 (the 8-GPU config): 8 shards sharing one position buffer on ONE GPU, the
    device-local stand-in for the RCCL all-gather, reproduce the unsharded run bit for bit."""
    import torch
    nw, nd, G, seed, P = 524288, 32, 6, 31, 8
    th = np.random.default_rng(17).standard_normal((nw, nd))
    pdf = kmc.GaussianIso()
    ref = _run(kmc, pdf, th, G, 2, seed)
    pos = torch.empty((nw, nd), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    shards = []
    for r in range(P):
        s = kmc.Sampler(pdf, nw, nd, G, 2, 1, 2.0, seed, moments=True, use_graph=False, shard_rank=r, shard_count=P)
        s.bind_positions(pos.data_ptr())
        s.set_stream(stream)
        s.set_positions(th)
        shards.append(s)
    for _ in range(G):
        for half in (0, 1):
            for s in shards:
                s.half_step(half)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pos.cpu().numpy(), ref["pos"])
    np.testing.assert_array_equal(sum(s.naccept() for s in shards), ref["nacc"])
    np.testing.assert_allclose(sum(s.moments()[0] for s in shards), ref["sum"], rtol=1e-11, atol=1e-8)
    for s in shards:
        s.close()


def test_c4_size_dealt_sub_ensembles_sample_the_target(kmc):
    """The C4 ensemble (8 x 65 536 walkers x 32-dim Gaussian) as dealt sub-ensembles -- 8 logical sub-ensembles on this one GPU,
    re-dealt every 64 generations (the opt-in multi-GPU mode without a per-half-step exchange; bit-identity with the oracle at
    smaller sizes: test_gpu_dealt.py).  Started at stationarity: acceptance 0.234 and posterior moments within 1 %, and walkers
    do travel between sub-ensembles."""
    from kissmcmc_jl_amd.distributed import HipDealExecutor, LocalDealtEmcee
    P, S, nd, G, nburn, E = 8, 65536, 32, 1200, 200, 64
    th = np.random.default_rng(9).standard_normal((P * S, nd))
    exs = [HipDealExecutor(kmc.GaussianIso(), S, nd, G, nburn, 1, 2.0, 2024, rank=r, world=P, device=0) for r in range(P)]
    drv = LocalDealtEmcee(exs, P * S, nd, E)
    try:
        drv.set_positions(th)
        drv.run(G)
        drv.sync()
        ids0 = exs[0].sampler.walker_ids()
        res = drv.results()
    finally:
        drv.close()
    assert res["n"] == P * S * (G - nburn)
    mean = res["sum"] / res["n"]
    var = res["sumsq"] / res["n"] - mean ** 2
    assert np.abs(mean).max() < 0.01 and np.abs(var - 1.0).max() < 0.01
    acc = res["naccept"] / (G - nburn)
    assert abs(acc.mean() - 0.234) < 0.00234 * 2 and acc.std() < 0.03
    origin = np.bincount(ids0 // S, minlength=P)                     # where sub-ensemble 0's walkers started
    assert origin.min() > S // P // 2                                # about S / P from each of the 8
    assert np.all(np.isfinite(res["positions"])) and len(np.unique(res["positions"][:, 0])) > 0.99 * P * S


def test_c2_chain_beyond_4g_elements(kmc):
    """A device-resident chain of 18.5 GB (C2, 1100 samples per walker: element offsets beyond 2^31, byte offsets beyond 2^33):
    stored samples equal the positions of independent runs stopped at the generations that produced them, the last sample
    is where the run ended, and the read-out by walker (transposed on the device in pieces) is the same data."""
    nw, nd, nthin, nsamp, seed = 65536, 32, 3, 1100, 2027
    try:                                                               # three 18.5 GB host arrays: only where the box's share allows
        limit = open("/sys/fs/cgroup/memory.max").read().strip()
        if limit != "max" and int(limit) < 100 * 2 ** 30:
            pytest.skip("needs ~60 GB of host memory")
    except OSError:
        pass
    G = nthin * nsamp
    th = np.random.default_rng(11).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 0, nthin, 2.0, seed, store_chain=True, store_logp=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        final, flogp = s.positions(), s.logp()
        ch, lp = s.chain()
        assert ch.shape == (nsamp, nw, nd) and ch.size > 2 ** 31 and ch.nbytes > 2 ** 33
        np.testing.assert_array_equal(ch[-1], final)
        np.testing.assert_array_equal(lp[-1], flogp)
        bw, lw = s.chain(by_walker=True)
        for k in (0, 1, 127, 128, 255, 256, 511, 777, 1023, 1024, 1099):   # byte offsets 0 ... 18.4 GB
            np.testing.assert_array_equal(bw[:, k], ch[k])
            np.testing.assert_array_equal(lw[:, k], lp[k])
        for w in (0, 1, 32767, 32768, 65535):
            np.testing.assert_array_equal(bw[w], ch[:, w])
        del bw, lw
    for k in (0, 130, 640, 1098):                                       # independent runs up to the generation of sample k
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 0, nthin, 2.0, seed) as t:
            t.set_positions(th)
            t.run((k + 1) * nthin)
            t.sync()
            np.testing.assert_array_equal(t.positions(), ch[k])
            np.testing.assert_array_equal(t.logp(), lp[k])
