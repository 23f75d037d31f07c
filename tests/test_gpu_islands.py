"""GPU: ISLAND MODE (opt-in extension): 256-walker islands resident in LDS, partners drawn inside the
island, walkers re-dealt between epochs.  Same target distribution as the reference's sampler, a
different partner pool -- so the checker is the oracle's own island restatement (bit-exact), plus
the analytic moments / acceptance rate the exact mode is held to."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(kmc, pdf, th, G, nburn, nthin, seed, k):
    nw, nd = th.shape
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, moments=True, island_gens=k) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        m = s.moments()
        return dict(pos=s.positions(), logp=s.logp(), nacc=s.naccept(), sum=m[0], sumsq=m[1], n=m[2],
                    ms=s.last_run_ms(), launches=s.launch_count)


CASES = [
    ("gauss", 512, 32, 70, 20, 1, 16),
    ("gauss", 1024, 32, 45, 10, 2, 7),       # epochs that do not divide the run
    ("gauss", 256, 5, 60, 20, 1, 8),         # ragged row
    ("gauss_shift", 768, 1, 80, 30, 1, 10),
    ("expo", 512, 3, 80, 20, 1, 16),
    ("rosen", 512, 16, 60, 20, 1, 12),
    ("rosen", 256, 2, 90, 30, 3, 32),
    ("lognormal", 256, 4, 60, 20, 1, 9),
]


def _dens(kmc, oracle, name):
    return {"gauss": (kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0]),
            "gauss_shift": (kmc.GaussianIso(-5.0, 3.0), oracle.GAUSSIAN_ISO, [-5.0, 3.0]),
            "expo": (kmc.Exponential(), oracle.EXPONENTIAL, [1.0]),
            "rosen": (kmc.Rosenbrock(), oracle.ROSENBROCK, [1.0, 100.0, 20.0]),
            "lognormal": (kmc.LogNormal(0.0, 1.0), oracle.LOGNORMAL, [0.0, 1.0])}[name]


@pytest.mark.parametrize("name,nw,nd,G,nburn,nthin,k", CASES)
def test_island_mode_matches_island_oracle(kmc, oracle, name, nw, nd, G, nburn, nthin, k):
    pdf, did, params = _dens(kmc, oracle, name)
    rng = np.random.default_rng(nd + nw)
    th = rng.standard_normal((nw, nd))
    if name in ("expo", "lognormal"):
        th = 0.5 + np.abs(th) * 0.2
    if name == "gauss_shift":
        th = -4.0 + 0.1 * th
    if name == "rosen":
        th *= 0.1
    got = _run(kmc, pdf, th, G, nburn, nthin, 31, k)
    cfg = oracle.make_config(did, params, nw, nd, G, nburn, nthin, 2.0, 31, nthreads=4)
    ref = oracle.emcee_islands(cfg, 256, k, th)
    assert ref["status"] == 0
    np.testing.assert_array_equal(got["nacc"], ref["naccept"])
    np.testing.assert_array_equal(got["pos"], ref["final_pos"])
    assert np.all(np.abs(got["logp"] - ref["final_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"])))
    assert got["n"] == ref["nmoment"]
    np.testing.assert_allclose(got["sum"], ref["sum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(got["sumsq"], ref["sumsq"], rtol=1e-11, atol=1e-9)


def test_island_mode_pieces_equal_one_run(kmc):
    th = np.random.default_rng(2).standard_normal((1024, 32))
    a = _run(kmc, kmc.GaussianIso(), th, 64, 16, 1, 5, 16)
    with kmc.Sampler(kmc.GaussianIso(), 1024, 32, 64, 16, 1, 2.0, 5, moments=True, island_gens=16) as s:
        s.set_positions(th)
        for n in (5, 11, 1, 30, 17):
            s.run(n)
        s.sync()
        np.testing.assert_array_equal(s.positions(), a["pos"])
        np.testing.assert_array_equal(s.naccept(), a["nacc"])
        np.testing.assert_allclose(s.moments()[0], a["sum"], rtol=1e-12, atol=1e-9)


def test_island_mode_c2_moments_acceptance_and_speed(kmc):
    """C2 shape, 10^4 generations: analytic posterior moments and acceptance within 1 %, like the exact
    mode (tests/test_gpu_fullsize.py); also reports the throughput of this mode."""
    th = np.random.default_rng(7).standard_normal((65536, 32))
    G, nburn = 10000, 5000
    got = _run(kmc, kmc.GaussianIso(), th, G, nburn, 1, 2024, 50)
    mean = got["sum"] / got["n"]
    var = got["sumsq"] / got["n"] - mean ** 2
    assert got["n"] == 65536 * (G - nburn)
    assert np.all(np.abs(mean) < 0.01) and np.all(np.abs(var - 1.0) < 0.01), (np.abs(mean).max(), var.min(), var.max())
    acc = got["nacc"] / (G - nburn)
    assert abs(acc.mean() - 0.234) < 0.00234 * 2
    assert abs(got["pos"].mean()) < 0.01 and abs(got["pos"].var() - 1.0) < 0.01
    print(f"island mode C2: {65536 * G / (got['ms'] * 1e-3) / 1e9:.2f} Gsteps/s, {got['launches']} launches")


def test_island_mode_rejects_unsupported_configs(kmc):
    with pytest.raises(kmc.KmcError, match="KMC_ISLANDS needs"):
        kmc.Sampler(kmc.GaussianIso(), 1000, 32, 10, island_gens=8)          # not a multiple of 256
    with pytest.raises(kmc.KmcError, match="KMC_ISLANDS needs"):
        kmc.Sampler(kmc.GaussianIso(), 1024, 64, 10, island_gens=8)          # ndim > 32
    with pytest.raises(kmc.KmcError, match="KMC_ISLANDS needs"):
        kmc.Sampler(kmc.GaussianIso(), 1024, 32, 10, island_gens=8, store_chain=True)


@pytest.mark.parametrize("S", [64, 128])
def test_island_mode_with_a_runtime_compiled_density(kmc, oracle, S):
    """ExprDensity in island mode (hiprtc-compiled island kernel): same chains as the menu density's
    island oracle."""
    nw, nd, G, nburn, k = 1024, 16, 60, 20, 12
    th = 0.1 * np.random.default_rng(4).standard_normal((nw, nd))
    pdf = kmc.ExprDensity("d < n-1 ? -((p[0]-x)*(p[0]-x))/p[2] : 0.0", "-(p[1]*((y-x*x)*(y-x*x)))/p[2]", [1.0, 100.0, 20.0])
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, 77, moments=True, island_gens=k, island_size=S) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        pos, nacc, mom = s.positions(), s.naccept(), s.moments()
    cfg = oracle.make_config(oracle.ROSENBROCK, [1.0, 100.0, 20.0], nw, nd, G, nburn, 1, 2.0, 77, nthreads=4)
    ref = oracle.emcee_islands(cfg, S, k, th)
    np.testing.assert_array_equal(nacc, ref["naccept"])
    np.testing.assert_array_equal(pos, ref["final_pos"])
    assert mom[2] == ref["nmoment"]
    np.testing.assert_allclose(mom[0], ref["sum"], rtol=1e-11, atol=1e-9)


def test_island_mode_with_a_user_density_at_the_largest_island(kmc):
    """A term / pair density in island mode with 256-walker islands of 32-dimensional rows (146 KiB of LDS -- runtime-compiled
    kernels used to be held to 60 KiB): the menu density's run, bit for bit."""
    nd, S = 32, 256
    out = {}
    for name, pdf in (("menu", kmc.GaussianIso()), ("expr", kmc.ExprDensity("-0.5*x*x"))):
        with kmc.Sampler(pdf, 4096, nd, 128, 32, 1, 2.0, 5, moments=True, island_gens=32, island_size=S) as s:
            assert "island mode" in s.describe()
            s.set_positions(np.random.default_rng(1).standard_normal((4096, nd)))
            s.run(128)
            s.sync()
            out[name] = (s.positions(), s.naccept())
    np.testing.assert_array_equal(out["menu"][0], out["expr"][0])
    np.testing.assert_array_equal(out["menu"][1], out["expr"][1])
