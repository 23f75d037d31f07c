"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp64):
  * accept decisions / naccept: identical
  * positions: bit-identical (the move is one fma per element on both sides; -ffp-contract=off)
  * log-pdf: |gpu - oracle| <= 1e-12 * max(1, |oracle|)  (summation order differs in the
    lane-striped kernel; libm vs device log differ by <= 1 ulp)
"""
import numpy as np
import pytest

import kmcenv

pytestmark = pytest.mark.gpu

LOGP_RTOL = 1e-12


def _densities(kmc, oracle):
    return {
        "gauss": (kmc.GaussianIso(0.0, 1.0), oracle.GAUSSIAN_ISO, [0.0, 1.0]),
        "gauss_shift": (kmc.GaussianIso(-5.0, 3.0), oracle.GAUSSIAN_ISO, [-5.0, 3.0]),
        "expo": (kmc.Exponential(1.0), oracle.EXPONENTIAL, [1.0]),
        "rosen": (kmc.Rosenbrock(1.0, 100.0, 20.0), oracle.ROSENBROCK, [1.0, 100.0, 20.0]),
        "lognormal": (kmc.LogNormal(0.0, 1.0), oracle.LOGNORMAL, [0.0, 1.0]),
    }


def _theta0(name, nw, nd, seed):
    rng = np.random.default_rng(seed)
    if name in ("expo", "lognormal"):
        return 0.5 + 0.1 * np.abs(rng.standard_normal((nw, nd))) + 0.05
    if name == "gauss_shift":
        return -4.0 + 0.1 * rng.standard_normal((nw, nd))
    return 0.1 * rng.standard_normal((nw, nd))


def _run_both(kmc, oracle, name, nw, nd, G, nburn, nthin, seed, use_graph=True, plan=None, monkeypatch=None):
    pdf, did, params = _densities(kmc, oracle)[name]
    th = _theta0(name, nw, nd, seed)
    if plan is not None:
        monkeypatch.setenv("KMC_PLAN", plan)
    cfg = oracle.make_config(did, params, nw, nd, G, nburn, nthin, 2.0, seed)
    ref = oracle.emcee(cfg, th)
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True,
                     moments=True, use_graph=use_graph) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(),
                   accept_ratio=s.accept_ratio())
        got["chain"], got["chain_logp"] = s.chain()
        got["sum"], got["sumsq"], got["nmoment"] = s.moments()
    return ref, got


def _compare(ref, got):
    assert ref["status"] == 0
    np.testing.assert_array_equal(got["naccept"], ref["naccept"])
    np.testing.assert_array_equal(got["final_pos"], ref["final_pos"])
    scale = np.maximum(1.0, np.abs(ref["final_logp"]))
    assert np.all(np.abs(got["final_logp"] - ref["final_logp"]) <= LOGP_RTOL * scale)
    assert got["chain"].shape == ref["chain"].shape
    np.testing.assert_array_equal(got["chain"], ref["chain"])
    scale = np.maximum(1.0, np.abs(ref["chain_logp"]))
    assert np.all(np.abs(got["chain_logp"] - ref["chain_logp"]) <= LOGP_RTOL * scale)
    np.testing.assert_allclose(got["accept_ratio"], ref["accept_ratio"], rtol=0, atol=0)
    assert got["nmoment"] == ref["nmoment"]
    np.testing.assert_allclose(got["sum"], ref["sum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(got["sumsq"], ref["sumsq"], rtol=1e-11, atol=1e-9)


CASES = [
    # name, nwalkers, ndim, G, nburnin, nthin
    ("gauss", 64, 4, 50, 10, 1),        # generic kernel
    ("gauss", 256, 32, 40, 10, 1),      # vector kernel L=16 K=1
    ("gauss", 130, 32, 30, 5, 2),       # ragged: active half not a multiple of the group count
    ("gauss", 256, 64, 20, 4, 1),       # L=32
    ("gauss", 1040, 1024, 6, 2, 1),     # L=64 K=8
    ("gauss_shift", 100, 1, 200, 100, 1),
    ("expo", 100, 1, 300, 150, 1),      # README shape (C1)
    ("expo", 128, 16, 40, 10, 3),
    ("rosen", 100, 2, 300, 100, 1),     # the reference's own Rosenbrock test shape
    ("rosen", 256, 64, 30, 10, 1),      # chained Rosenbrock on the vector kernel (C3 shape)
    ("rosen", 2100, 1024, 4, 1, 1),     # K > 1: neighbour wraps across chunks
    ("lognormal", 100, 1, 200, 100, 1),
    # ragged sizes: ndim not a power of two / odd (masked last chunk, padded row stride)
    ("gauss", 200, 3, 60, 20, 1),
    ("gauss_shift", 256, 5, 60, 20, 2),
    ("expo", 128, 7, 50, 10, 1),
    ("rosen", 128, 10, 80, 20, 1),
    ("rosen", 130, 9, 80, 20, 1),
    ("gauss", 300, 33, 40, 10, 1),
    ("rosen", 256, 50, 40, 10, 1),
    ("lognormal", 128, 6, 60, 20, 1),
    ("gauss", 512, 100, 30, 10, 1),
    ("gauss_shift", 600, 200, 20, 5, 1),
    ("rosen", 700, 300, 10, 3, 1),
    ("expo", 1100, 1000, 6, 2, 1),
    ("gauss", 1100, 1023, 6, 2, 1),
    ("gauss", 1040, 1026, 4, 1, 1),     # beyond the vector kernel's 1024: generic kernel
    # smallest legal ensembles: nwalkers == ndim + 2 (src/samplers.jl:205), two walkers per half
    ("gauss", 4, 2, 300, 100, 1),
    ("expo", 4, 1, 300, 100, 1),
    ("rosen", 6, 4, 200, 50, 3),
    ("gauss", 34, 32, 100, 30, 1),
]


@pytest.mark.parametrize("name,nw,nd,G,nburn,nthin", CASES)
def test_matches_oracle(kmc, oracle, name, nw, nd, G, nburn, nthin):
    ref, got = _run_both(kmc, oracle, name, nw, nd, G, nburn, nthin, seed=1234 + nd)
    _compare(ref, got)


@pytest.mark.parametrize("name,nw,nd,plan", [("gauss", 2048, 128, "32,2,4"), ("gauss", 4100, 128, "32,2,8"), ("rosen", 1030, 100, "32,2,4"), ("gauss_shift", 2048, 32, "8,2,8")])
def test_moments_folded_per_workgroup(kmc, oracle, name, nw, nd, plan, monkeypatch):
    """ITER >= 4 (the planner's geometries for ensembles that live in HBM) with several waves per workgroup: the waves' moment sums are added up through
    LDS and one wave per workgroup rewrites its accumulator slots (kmc_kernels.hpp: kWgFold) -- same chain, moments to 1e-11, partly idle last workgroups."""
    ref, got = _run_both(kmc, oracle, name, nw, nd, 70, 20, 3, seed=7, plan=plan, monkeypatch=monkeypatch)
    _compare(ref, got)


@pytest.mark.parametrize("plan", ["generic", "16,1,1", "16,1,2", "16,1,4", "16,1,8", "16,1,16", "8,2,1", "8,2,4", "8,2,8", "4,4,1", "4,4,4", "4,2,2"])
def test_every_geometry_gives_the_same_chain(kmc, oracle, plan, monkeypatch):
    """The result is a pure function of (seed, inputs): launch geometry must not matter."""
    ref, got = _run_both(kmc, oracle, "gauss", 512, 32, 70, 20, 1, seed=99, plan=plan, monkeypatch=monkeypatch)
    _compare(ref, got)


@pytest.mark.parametrize("a_scale", [1.0001, 1.5, 3.5, 10.0])
def test_stretch_scale(kmc, oracle, a_scale):
    """a_scale only needs to be > 1 (src/samplers.jl:200); z in [1/a, a]."""
    pdf, did, params = _densities(kmc, oracle)["gauss"]
    th = _theta0("gauss", 128, 8, 3)
    ref = oracle.emcee(oracle.make_config(did, params, 128, 8, 90, 20, 1, a_scale, 77), th)
    with kmc.Sampler(pdf, 128, 8, 90, 20, 1, a_scale, 77, store_chain=True, store_logp=True, moments=True) as s:
        s.set_positions(th)
        s.run(90)
        s.sync()
        got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio())
        got["chain"], got["chain_logp"] = s.chain()
        got["sum"], got["sumsq"], got["nmoment"] = s.moments()
    _compare(ref, got)


def test_run_in_pieces_equals_one_run(kmc, oracle):
    """Progress-meter style chunked runs, a mid-run moments read-out and a restart from the same
    positions all leave the chain unchanged."""
    pdf, did, params = _densities(kmc, oracle)["gauss"]
    th = _theta0("gauss", 256, 32, 4)
    ref = oracle.emcee(oracle.make_config(did, params, 256, 32, 200, 60, 2, 2.0, 5), th)
    with kmc.Sampler(pdf, 256, 32, 200, 60, 2, 2.0, 5, store_chain=True, store_logp=True, moments=True) as s:
        for attempt in range(2):
            s.set_positions(th)
            for n in (1, 63, 64, 7, 65):
                s.run(n)
                s.sync()
                s.moments()              # flushes the sojourn-weighted accumulators mid-run
            assert s.generation == 200
            got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio())
            got["chain"], got["chain_logp"] = s.chain()
            got["sum"], got["sumsq"], got["nmoment"] = s.moments()
            _compare(ref, got)


def test_graph_replay_equals_eager(kmc, oracle):
    """A run long enough to replay the hipGraph twice plus an eager tail vs eager launches only."""
    ref, got = _run_both(kmc, oracle, "gauss", 256, 32, 150, 70, 3, seed=5)
    _compare(ref, got)
    ref2, got2 = _run_both(kmc, oracle, "gauss", 256, 32, 150, 70, 3, seed=5, use_graph=False)
    _compare(ref2, got2)
    np.testing.assert_array_equal(got["chain"], got2["chain"])


@pytest.mark.parametrize("name,nw,nd", [("expo", 100, 1), ("rosen", 100, 2), ("gauss", 256, 32), ("gauss", 34, 32), ("lognormal", 64, 3),
                                        ("gauss", 512, 32), ("rosen", 400, 16), ("expo", 1024, 8), ("gauss", 1000, 3)])
def test_resident_small_ensemble_kernel_is_the_same_sampler(kmc, oracle, name, nw, nd, monkeypatch, kmc_debug):
    """nwalkers <= 1024: the whole ensemble runs out of one workgroup's LDS, many generations per launch
    (resident mode) -- one walker per thread for short rows (ndim <= 8), two lanes per walker otherwise; KMC_DEBUG=resident=pair
    keeps the two-lane kernel on short rows as well.  Each must be indistinguishable from the launch-per-half-step kernels
    and the oracle."""
    ref, res = _run_both(kmc, oracle, name, nw, nd, 200, 60, 2, seed=321)
    _compare(ref, res)
    if nd <= 8:
        kmc_debug.set("resident", "pair")
        with kmc.Sampler(_densities(kmc, oracle)[name][0], nw, nd, 10) as s:
            assert "resident mode" in s.describe() and "2 lanes" in s.describe()
        ref1, pair = _run_both(kmc, oracle, name, nw, nd, 200, 60, 2, seed=321)
        _compare(ref1, pair)
        np.testing.assert_array_equal(res["chain"], pair["chain"])
        kmc_debug.unset("resident")
        with kmc.Sampler(_densities(kmc, oracle)[name][0], nw, nd, 10) as s:
            assert "one walker per thread" in s.describe()
    kmcenv.no_resident(monkeypatch)
    ref2, multi = _run_both(kmc, oracle, name, nw, nd, 200, 60, 2, seed=321)
    _compare(ref2, multi)
    np.testing.assert_array_equal(res["chain"], multi["chain"])
    np.testing.assert_array_equal(res["naccept"], multi["naccept"])


@pytest.mark.parametrize("name,nw,nd,G,nburn,nthin,pieces", [
    ("expo", 100, 1, 2500, 1100, 7, None),                 # three launches (1024 generations each at most); burn-in ends inside the second
    ("gauss", 300, 5, 2100, 1024, 3, (1, 1022, 3, 1074)),  # run() pieces that end one short of a launch / a batch of draws
    ("gauss", 100, 20, 1500, 700, 5, (1023, 2, 475)),      # two lanes per walker
    ("rosen", 64, 2, 3000, 0, 1000, (7, 2993)),            # no burn-in, thinning longer than a launch
])
def test_resident_runs_longer_than_a_launch_have_no_seam(kmc, oracle, name, nw, nd, G, nburn, nthin, pieces):
    """A resident launch carries at most 1024 generations and its own table of draws (batches of four generations); the thinning
    phase is carried, not recomputed.  Launch boundaries, run() calls and batch tails must not show: the oracle's uninterrupted run."""
    pdf, did, params = _densities(kmc, oracle)[name]
    th = _theta0(name, nw, nd, 9)
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, nthin, 2.0, 77), th)
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, 77, store_chain=True, store_logp=True, moments=True) as s:
        assert "resident mode" in s.describe()
        s.set_positions(th)
        for n in (pieces or (G,)):
            s.run(n)
        s.sync()
        assert s.generation == G
        got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio())
        got["chain"], got["chain_logp"] = s.chain()
        got["sum"], got["sumsq"], got["nmoment"] = s.moments()
    _compare(ref, got)


@pytest.mark.parametrize("name,nw,nd,G,nburn,nthin", [("gauss", 1026, 4, 150, 40, 3), ("expo", 2048, 1, 120, 0, 1), ("rosen", 1500, 2, 200, 60, 2),
                                                    ("lognormal", 1100, 7, 100, 30, 1), ("gauss_shift", 2000, 5, 90, 10, 4), ("gauss", 1990, 8, 60, 20, 1),
                                                    ("gauss", 1300, 3, 2100, 1000, 50)])
def test_resident_two_walkers_per_thread(kmc, oracle, name, nw, nd, G, nburn, nthin, monkeypatch):
    """1026 .. 2048 walkers with short rows (as LDS allows): resident mode with two walkers per thread -- the oracle's chain, and the
    multi-launch kernels' to the last bit (an ensemble of 1100 walkers used to run 4.7x slower than one of 1000)."""
    ref, res = _run_both(kmc, oracle, name, nw, nd, G, nburn, nthin, seed=99)
    _compare(ref, res)
    with kmc.Sampler(_densities(kmc, oracle)[name][0], nw, nd, 10) as s:
        fits = nw * ((nd | 1) + 1) * 8 <= 156 * 1024
        assert ("two walkers per thread" in s.describe()) == fits, s.describe()
    kmcenv.no_resident(monkeypatch)
    ref2, multi = _run_both(kmc, oracle, name, nw, nd, G, nburn, nthin, seed=99)
    np.testing.assert_array_equal(res["chain"], multi["chain"])
    np.testing.assert_array_equal(res["naccept"], multi["naccept"])
    np.testing.assert_array_equal(res["final_pos"], multi["final_pos"])


def test_describe_reports_the_execution_mode(kmc):
    with kmc.Sampler(kmc.GaussianIso(), 65536, 32, 10) as s:
        assert "half_step_vec L=8 K=2 ITER=2 exact-size" in s.describe() and "hipGraph" in s.describe()
    with kmc.Sampler(kmc.GaussianIso(), 100, 1, 10) as s:
        assert "resident mode" in s.describe()
    with kmc.Sampler(kmc.GaussianIso(), 4096, 10, 10, use_graph=False) as s:
        assert "ragged" in s.describe() and "eager" in s.describe()
    with kmc.Sampler(kmc.GaussianIso(), 4096, 1030, 10) as s:
        assert "half_step_generic" in s.describe()
    with kmc.Sampler(kmc.GaussianIso(), 1024, 8, 10, island_gens=4) as s:
        assert "island mode: 4 islands of 256" in repr(s)


@pytest.mark.parametrize("mode", ["graph", "eager", "updated", None])
def test_launch_modes_are_the_same_sampler(kmc, oracle, mode, monkeypatch):
    """kmc_sampler_run issues the half-steps as a table-driven hipGraph, as eager launches, or as a graph whose node
    parameters are rewritten before every replay (KMC_LAUNCH; unset: a long run measures and picks): the kernels take
    their generation from a device table in the first case and from preloaded kernel parameters in the others --
    same draws, same result, and a chunked run crosses every seam (calibration, replay, eager tail)."""
    if mode is None:
        monkeypatch.delenv("KMC_LAUNCH", raising=False)
    else:
        monkeypatch.setenv("KMC_LAUNCH", mode)
    monkeypatch.setenv("KMC_DEBUG", "fused=0")      # the launch modes of the two-launch kernels (this small state would run one launch per generation)
    nw, nd, G, nburn, nthin, seed = 2048, 32, 1000, 301, 7, 23
    th = np.random.default_rng(4).standard_normal((nw, nd))
    planned = 4200 if mode is None else G          # (not forced: a job PLANNED long -- >= 4096 generations -- measures the modes at its first call of >= 896; 1000 of them run here)
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, planned, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
        s.set_positions(th)
        s.run(900)          # long enough for the one-off measurement when the mode is not forced
        s.run(37)
        s.run(63)
        s.sync()
        pos, nacc = s.positions(), s.naccept()
        chain, clogp = s.chain()
        msum, msq, n = s.moments()
        how = s.describe()
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, nthin, 2.0, seed, nthreads=8), th)
    np.testing.assert_array_equal(nacc, ref["naccept"])
    np.testing.assert_array_equal(pos, ref["final_pos"])
    np.testing.assert_array_equal(chain, ref["chain"])
    assert n == ref["nmoment"]
    np.testing.assert_allclose(msum, ref["sum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(msq, ref["sumsq"], rtol=1e-11, atol=1e-9)
    if mode == "updated":
        assert "parameter updates" in how
    if mode is None:
        assert "measured per 64 generations" in how, how
    if mode == "eager":
        assert "eager" in how


@pytest.mark.parametrize("when", ["spent_before_the_run", "spent_during_the_run"])
def test_updated_graph_budget_fallbacks_are_the_same_sampler(kmc, oracle, when, monkeypatch, capfd):
    """The updated-graph mode has a process-wide budget of parameter updates (the runtime keeps ~80 B per update -- HIP 7.0, PyTorch's -- or ~1.4 B -- 7.2).  A sampler
    that finds it spent -- before its first long run, or in the middle of one -- goes on with the table graph or eager
    launches: same results as the oracle, and it SAYS so (stderr once per process, describe(), kmc_sampler_launch_mode)."""
    import ctypes as C
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    used, budget = C.c_int64(0), C.c_int64(0)
    L.kmc_updated_budget(C.byref(used), C.byref(budget))
    L.kmc_set_updated_budget_mb(1.0)                   # what the library prices an update at: 80 B on the HIP 7.0 runtime (PyTorch's), 2 B on 7.2
    one_mib = C.c_int64(0)
    L.kmc_updated_budget(None, C.byref(one_mib))
    each = 1048576.0 / one_mib.value
    assert abs(each - 80.0) < 0.01 or abs(each - 2.0) < 0.01, each
    each = float(round(each))
    nw, nd, G, nburn, nthin, seed = 2048, 32, 1000, 301, 7, 29
    th = np.random.default_rng(5).standard_normal((nw, nd))
    monkeypatch.setenv("KMC_DEBUG", "fused=0")      # (the two-launch kernels' launch modes: this small state would run one launch per generation)
    try:
        if when == "spent_before_the_run":
            monkeypatch.delenv("KMC_LAUNCH", raising=False)
            L.kmc_set_updated_budget_mb(13 * each / 1048576.0)              # 13 updates, fewer than one replay needs
        else:
            monkeypatch.setenv("KMC_LAUNCH", "updated,budget")              # in the updated-graph mode, budget applies
            L.kmc_set_updated_budget_mb((used.value + 3 * 256) * each / 1048576.0 + 1e-9)   # room for three replays (128 generations, two updates each)
        planned = 4200          # (a job planned long -- >= 4096 generations -- is what measures its launch modes, and so meets the budget; 1000 of them run here)
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, planned, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
            s.set_positions(th)
            s.run(900)
            s.run(37)
            s.run(63)
            s.sync()
            pos, nacc = s.positions(), s.naccept()
            chain, clogp = s.chain()
            msum, msq, n = s.moments()
            how = s.describe()
            mode, fell_back = s.launch_mode()
            assert s.generation == G
    finally:
        L.kmc_set_updated_budget_mb(budget.value * each / 1048576.0 + 1e-9)
    assert fell_back and mode in (0, 1, 2), (mode, fell_back, how)
    assert "budget of the process spent" in how
    u2, b2 = C.c_int64(0), C.c_int64(0)
    L.kmc_updated_budget(C.byref(u2), C.byref(b2))
    assert b2.value == budget.value
    if when == "spent_during_the_run":
        assert 3 * 256 <= u2.value - used.value <= 4 * 256               # it stopped updating when the budget ran out
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, nthin, 2.0, seed, nthreads=8), th)
    np.testing.assert_array_equal(nacc, ref["naccept"])
    np.testing.assert_array_equal(pos, ref["final_pos"])
    np.testing.assert_array_equal(chain, ref["chain"])
    assert n == ref["nmoment"]
    np.testing.assert_allclose(msum, ref["sum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(msq, ref["sumsq"], rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("depth", [None, 2, "off"])
def test_moment_ring_of_long_rows(kmc, oracle, depth, monkeypatch, kmc_debug):
    """ndim > 256: waves with an accepted move post the replaced row into a per-wave ring and moments_sweep folds the
    entries between graph chunks (HalfStepArgs::mring).  Same sums with the default depth, with a two-entry ring that
    overflows into the read-modify-write path all the time, and without the ring; 200 generations cross three graph
    chunks, an eager tail and a read-out in the middle."""
    if depth == "off":
        kmc_debug.set("no-moment-ring")
    elif depth is not None:
        kmc_debug.set("moment-ring-depth", str(depth))
    nw, nd, G, nburn, nthin, seed = 1040, 600, 200, 20, 1, 5
    th = np.random.default_rng(3).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, nthin, 2.0, seed, moments=True) as s:
        s.set_positions(th)
        s.run(130)
        s.sync()
        mid = s.moments()
        s.run(70)
        s.sync()
        msum, msq, n = s.moments()
        nacc = s.naccept()
        assert ("ring of" in s.describe()) == (depth != "off")
    cfg = lambda g: oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, g, nburn, nthin, 2.0, seed, nthreads=8)
    ref_mid = oracle.emcee(cfg(130), th, store_chain=False)
    ref = oracle.emcee(cfg(G), th, store_chain=False)
    np.testing.assert_array_equal(nacc, ref["naccept"])
    assert mid[2] == ref_mid["nmoment"] and n == ref["nmoment"]
    np.testing.assert_allclose(mid[0], ref_mid["sum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(mid[1], ref_mid["sumsq"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(msum, ref["sum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(msq, ref["sumsq"], rtol=1e-11, atol=1e-9)
