"""GPU: the KMC_F32 option (include/kissmcmc_hip.h) -- walker rows and the stored chain kept in IEEE single on the
device, arithmetic in double -- against the oracle run with ``state_f32`` (the initial ensemble and every proposal
rounded to float before the log-density is evaluated).  Same bar as the double path: identical accept decisions,
bit-identical positions and chain, log-pdfs within 1e-12 relative; and the reference's statistical pins."""
import ctypes as C

import numpy as np
import pytest

import kmcenv

from test_gpu_parity import _compare, _densities, _theta0

pytestmark = pytest.mark.gpu

CASES = [
    # name, nwalkers, ndim, G, nburnin, nthin
    ("gauss", 64, 4, 50, 10, 1),
    ("gauss", 4096, 32, 30, 10, 1),      # the C2 geometry (L=8 K=2, two walkers per group)
    ("gauss", 130, 32, 30, 5, 2),        # partial last wave
    ("gauss", 256, 64, 20, 4, 1),
    ("gauss", 1040, 1024, 6, 2, 1),      # L=64 K=8
    ("gauss_shift", 100, 1, 200, 100, 1),
    ("expo", 100, 1, 300, 150, 1),       # README shape
    ("rosen", 100, 2, 300, 100, 1),
    ("rosen", 2048, 64, 30, 10, 1),      # C3 geometry
    ("lognormal", 128, 6, 60, 20, 1),
    ("gauss", 300, 33, 40, 10, 1),       # odd ndim: padded float rows
    ("rosen", 130, 9, 80, 20, 1),
    ("gauss", 1040, 1026, 4, 1, 1),      # beyond the vector kernels: generic kernel
    ("gauss", 34, 32, 100, 30, 1),
    ("gauss", 1000, 4, 1100, 500, 7),    # resident mode, one walker per thread, more than one launch (draw table refilled)
    ("expo", 256, 3, 200, 50, 3),
]


def _run_f32(kmc, oracle, name, nw, nd, G, nburn, nthin, seed, use_graph=True):
    pdf, did, params = _densities(kmc, oracle)[name]
    th = _theta0(name, nw, nd, seed)
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, nthin, 2.0, seed, state_f32=True), th)
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, moments=True,
                     use_graph=use_graph, dtype="f32") as s:
        assert "float" in s.describe()
        s.set_positions(th)
        s.run(G)
        s.sync()
        got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio(), how=s.describe())
        got["chain"], got["chain_logp"] = s.chain()
        got["sum"], got["sumsq"], got["nmoment"] = s.moments()
    return ref, got, th


@pytest.mark.parametrize("name,nw,nd,G,nburn,nthin", CASES)
def test_f32_rows_match_the_oracle(kmc, oracle, name, nw, nd, G, nburn, nthin, monkeypatch):
    ref, got, th = _run_f32(kmc, oracle, name, nw, nd, G, nburn, nthin, seed=77 + nd)
    _compare(ref, got)
    if nw <= 1024 and nd <= 8:
        # short rows of a small ensemble run out of one workgroup's LDS (float rows widened on the way in, proposals rounded
        # before their density, as everywhere); the multi-launch kernels on the same job must agree too
        assert "resident mode" in got["how"] and "one walker per thread" in got["how"], got["how"]
        kmcenv.no_resident(monkeypatch)
        ref2, got2, _ = _run_f32(kmc, oracle, name, nw, nd, G, nburn, nthin, seed=77 + nd)
        assert "multi-launch" in got2["how"], got2["how"]
        _compare(ref2, got2)
        kmcenv.resident_again(monkeypatch)
    else:
        assert "resident" not in got["how"], got["how"]
    # everything stored is representable in single, and the run differs from the double one
    assert np.array_equal(got["chain"], got["chain"].astype(np.float32).astype(np.float64))
    assert np.array_equal(got["final_pos"], got["final_pos"].astype(np.float32).astype(np.float64))


def test_f32_eager_and_state_round_trip(kmc, oracle):
    ref, got, th = _run_f32(kmc, oracle, "gauss", 512, 32, 24, 4, 1, seed=5, use_graph=False)
    _compare(ref, got)
    pdf = kmc.GaussianIso()
    with kmc.Sampler(pdf, 512, 32, 24, 4, 1, 2.0, 5, dtype="f32") as a, kmc.Sampler(pdf, 512, 32, 24, 4, 1, 2.0, 5, dtype="f32") as b:
        a.set_positions(th)
        a.run(10)
        a.sync()
        b.restore(a.state())
        b.run(14)
        b.sync()
        np.testing.assert_array_equal(b.positions(), ref["final_pos"])
        np.testing.assert_array_equal(b.naccept(), ref["naccept"])


def test_f32_init_ball_and_reference_pins(kmc):
    """the reference's Normal(-5,3) case (test/runtests.jl:53-56) through emcee(..., dtype="f32")."""
    from refcases import CASES as REF, check_mean_std
    case = REF[0]
    pdf = kmc.GaussianIso(-5.0, 3.0)
    theta0s = kmc.make_theta0s(-4.0, 0.1, pdf, 100, rng=2)
    thetas, ar, logd, blobs = kmc.emcee(pdf, theta0s, niter=10 ** 5, use_progress_meter=False, seed=4, dtype="f32")
    assert thetas.shape == (100, 500) and np.all(ar > 0.1)
    t, a, l, _ = kmc.squash_walkers(thetas, ar, logd, verbose=False)
    check_mean_std(t, case)
    np.testing.assert_allclose(l, [pdf(float(v)) for v in t], rtol=1e-12, atol=1e-12)   # logp of the rows as stored
    with kmc.Sampler(kmc.Exponential(), 256, 3, 10, dtype="f32") as s:
        s.init_ball(np.full(3, 0.05), np.full(3, 0.1), seed=9)
        x, lp = s.positions(), s.logp()
        assert np.all(x >= 0) and np.array_equal(x, x.astype(np.float32).astype(np.float64))
        np.testing.assert_allclose(lp, -x.sum(axis=1), rtol=1e-14)


def test_f32_statistics_at_c2_geometry(kmc):
    """moments and acceptance of a stationary Gaussian ensemble, float rows against double rows (same seed: the
    chains decorrelate after the first rounding difference, the statistics must agree)."""
    rng = np.random.default_rng(0)
    th = rng.standard_normal((8192, 32))
    out = {}
    for dt in ("f64", "f32"):
        with kmc.Sampler(kmc.GaussianIso(), 8192, 32, 400, 100, 1, 2.0, 3, moments=True, dtype=dt) as s:
            s.set_positions(th)
            s.run(400)
            s.sync()
            sm, sq, n = s.moments()
            out[dt] = (sm / n, sq / n - (sm / n) ** 2, s.accept_ratio().mean())
    for k in range(2):
        np.testing.assert_allclose(out["f32"][k], out["f64"][k], atol=0.02)
    assert abs(out["f32"][2] - out["f64"][2]) < 0.003
    assert np.all(np.abs(out["f32"][0]) < 0.03) and np.all(np.abs(out["f32"][1] - 1.0) < 0.05)


def test_f32_is_refused_where_it_is_not_built(kmc):
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    c = _lib.Config()
    c.dtype, c.density = _lib.F32, 0
    c.params[0], c.params[1] = 0.0, 1.0
    c.nwalkers, c.ndim, c.ngenerations, c.nthin, c.a_scale = 64, 4, 10, 1, 2.0
    assert L.kmc_validate(C.byref(c)) == _lib.OK
    c.flags = _lib.P2P
    assert L.kmc_validate(C.byref(c)) == _lib.ERR_UNSUPPORTED
    c.flags, c.shard_count = 0, 2
    assert L.kmc_validate(C.byref(c)) == _lib.ERR_UNSUPPORTED
    c.shard_count, c.dtype = 1, 7
    assert L.kmc_validate(C.byref(c)) == _lib.ERR_UNSUPPORTED
    with pytest.raises(_lib.KmcError, match="evaluated on the device"):
        kmc.Sampler(kmc.HostLogPdf(lambda x: 0.0), 64, 4, 10, dtype="f32")
    with kmc.Sampler(kmc.GaussianIso(), 64, 4, 10, dtype="f32") as s:
        with pytest.raises(_lib.KmcError, match="double rows"):
            s.bind_positions(1 << 20)


@pytest.mark.parametrize("nw,nd,plan", [(256, 32, ""), (200, 5, ""), (256, 64, ""), (256, 32, "generic")])
def test_f32_rows_with_runtime_compiled_densities(kmc, oracle, nw, nd, plan, monkeypatch):
    """ExprDensity restating a menu density, float rows: the oracle's state_f32 chain of that density."""
    if plan:
        monkeypatch.setenv("KMC_PLAN", plan)
    rosen = nd == 64
    pdf = (kmc.ExprDensity("d < n-1 ? -((p[0]-x)*(p[0]-x))/p[2] : 0.0", "-(p[1]*((y-x*x)*(y-x*x)))/p[2]", [1.0, 100.0, 20.0])
           if rosen else kmc.ExprDensity("-0.5*x*x"))
    did, params = (oracle.ROSENBROCK, [1.0, 100.0, 20.0]) if rosen else (oracle.GAUSSIAN_ISO, [0.0, 1.0])
    th = (0.1 if rosen else 1.0) * np.random.default_rng(nd).standard_normal((nw, nd))
    G, nburn, seed = 80, 20, 7
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, 1, 2.0, seed, state_f32=True), th)
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed, store_chain=True, store_logp=True, moments=True, dtype="f32") as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio(), how=s.describe())
        got["chain"], got["chain_logp"] = s.chain()
        got["sum"], got["sumsq"], got["nmoment"] = s.moments()
    _compare(ref, got)


def test_f32_body_density_runs_resident(kmc, oracle, monkeypatch):
    """A user-written function body on float rows: one walker per thread out of LDS, equal to the multi-launch kernels
    bit for bit (same rounding points)."""
    body = "double s = 0; for (int i = 0; i < n; ++i) { double d = x[i] - p[0]; s += d * d; } return -0.5 * s / (p[1] * p[1]);"
    nw, nd, G = 200, 5, 300
    th = np.random.default_rng(12).standard_normal((nw, nd)).astype(np.float32).astype(np.float64)

    def run():
        with kmc.Sampler(kmc.CDensity(body, params=[0.5, 2.0]), nw, nd, G, 100, 2, 2.0, 9, store_chain=True, store_logp=True,
                         moments=True, dtype="f32") as s:
            s.set_positions(th)
            s.run(G)
            s.sync()
            ch, cl = s.chain()
            return dict(how=s.describe(), pos=s.positions(), logp=s.logp(), nacc=s.naccept(), chain=ch, clogp=cl, mom=s.moments())

    res = run()
    assert "resident mode" in res["how"] and "float" in res["how"], res["how"]
    kmcenv.no_resident(monkeypatch)
    ml = run()
    assert "multi-launch" in ml["how"], ml["how"]
    for k in ("pos", "nacc", "chain"):
        np.testing.assert_array_equal(res[k], ml[k], err_msg=k)
    np.testing.assert_allclose(res["logp"], ml["logp"], rtol=1e-12)
    np.testing.assert_allclose(res["clogp"], ml["clogp"], rtol=1e-12)
    for a_, b_ in zip(res["mom"][:2], ml["mom"][:2]):
        np.testing.assert_allclose(a_, b_, rtol=1e-11)
    assert np.array_equal(res["chain"], res["chain"].astype(np.float32).astype(np.float64))
    assert abs(res["chain"].mean() - 0.5) < 0.25
