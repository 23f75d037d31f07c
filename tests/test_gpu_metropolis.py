"""GPU leg of the many-chain Metropolis path (reference src/samplers.jl:59-128): the HIP kernel through the
C ABI against the oracle and the committed golden fixtures.  Tolerance (written here, as the path is floating
point with libm calls): accept decisions and counters identical; states, chains and sums within 1e-11
(device log/sin/cos vs glibc differ by <= 1 ulp in the Box-Muller normals), log-pdfs within 1e-10."""
import numpy as np
import pytest

import refcases
from test_metropolis_cpu import compare_golden, golden_names, load_golden

pytestmark = pytest.mark.gpu

_DENS = {0: lambda k, p: k.GaussianIso(p[0], p[1]), 1: lambda k, p: k.Exponential(p[0]),
         2: lambda k, p: k.Rosenbrock(p[0], p[1], p[2]), 3: lambda k, p: k.LogNormal(p[0], p[1])}


def _pdf(kmc, z):
    if z["density"] == 4:
        P = np.array([[z["params"][2], z["params"][3]], [z["params"][3], z["params"][4]]])
        pdf = kmc.MvNormal2(z["params"][:2], np.linalg.inv(P))
        raw = [float(v) for v in z["params"][:5]]
        pdf.params = lambda: raw          # the fixture's precision matrix verbatim, not inv(inv(P))
        return pdf
    return _DENS[z["density"]](kmc, z["params"])


@pytest.fixture(params=["table", "kernel"])
def draws(request, monkeypatch, kmc_debug):
    """Where the chains' draws are made: by a wide kernel first, read from a table (few chains: the default up to 16 384), or
    in the chains' own loop (many chains).  Same stream, same results; the parity tests run both whatever the chain count."""
    kmc_debug.set("metro-table", "1" if request.param == "table" else "0")
    return request.param


@pytest.mark.parametrize("name", golden_names())
def test_kernel_reproduces_metropolis_golden(kmc, name, draws):
    from kissmcmc_jl_amd.metropolis import run_chains
    z = load_golden(name)
    r = run_chains(_pdf(kmc, z), kmc.GaussianStep(z["step"]), z["theta0"], z["niter"], z["nburnin"], z["nthin"], z["seed"],
                   moments=True)
    compare_golden(z, r, exact=False)
    np.testing.assert_array_equal(r["accept_ratio"], r["naccept"] / (z["niter"] - z["nburnin"]))    # samplers.jl:127


@pytest.mark.parametrize("ndim", [1, 2, 3, 5, 8, 13, 16, 17, 32, 33, 40])
def test_every_register_geometry_equals_the_oracle(kmc, oracle, ndim, draws):
    """ndim 1..32 run in registers (6 kernel geometries), beyond that from memory; odd sizes mask the tail."""
    from kissmcmc_jl_amd.metropolis import run_chains
    nc, niter, nburn, nthin, seed = 777, 70, 21, 3, 100 + ndim          # a ragged last workgroup
    th = 0.3 * np.random.default_rng(ndim).standard_normal((nc, ndim))
    step = np.linspace(0.2, 0.6, ndim)
    r = run_chains(kmc.GaussianIso(0.1, 1.3), kmc.GaussianStep(step), th, niter, nburn, nthin, seed, moments=True)
    ref = oracle.metropolis(oracle.GAUSSIAN_ISO, [0.1, 1.3], th, step, niter, nburn, nthin, seed)
    np.testing.assert_array_equal(r["naccept"], ref["naccept"])
    np.testing.assert_allclose(r["chain"], ref["chain"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(r["chain_logp"], ref["chain_logp"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(r["final_pos"], ref["final_pos"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(r["chain_sum"], ref["chain_sum"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(r["chain_sumsq"], ref["chain_sumsq"], rtol=1e-10, atol=1e-10)


def test_long_run_is_cut_into_launches_without_a_seam(kmc, oracle, draws):
    """More iterations than one launch carries (65 536): thinning phase, sample slots and counters continue."""
    from kissmcmc_jl_amd.metropolis import run_chains
    nc, niter, nburn, nthin = 64, 140001, 30000, 7
    th = np.random.default_rng(5).standard_normal((nc, 2))
    pdf = kmc.Rosenbrock()
    r = run_chains(pdf, kmc.GaussianStep(0.5), th, niter, nburn, nthin, 77, moments=True)
    ref = oracle.metropolis(oracle.ROSENBROCK, [1.0, 100.0, 20.0], th, 0.5, niter, nburn, nthin, 77, nthreads=8)
    assert r["chain"].shape == ((niter - nburn) // nthin, nc, 2)
    np.testing.assert_array_equal(r["naccept"], ref["naccept"])
    np.testing.assert_allclose(r["chain"], ref["chain"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("case", [c for c in refcases.CASES if c["niter"] <= 10 ** 5], ids=lambda c: c["name"])
def test_reference_metropolis_cases_drop_in(kmc, case):
    """reference test/metro.jl:2-21 through the reference's own signature (one chain = one lane)."""
    if case["dens"] == "gauss":
        pdf = kmc.GaussianIso(*case["params"])
    else:
        pdf = kmc.MvNormal2(case["params"]["mean"], case["params"]["cov"])
    thetas, accept_ratio, logdensities, blobs = kmc.metropolis(pdf, kmc.GaussianStep(case["mstep"]), case["theta0"],
                                                               niter=case["niter"], use_progress_meter=False, seed=31)
    assert blobs is None                                                    # metro.jl:12
    assert len(thetas) == case["niter"] // 2                                # metro.jl:13
    assert len(logdensities) == case["niter"] // 2                          # metro.jl:14
    assert 0.15 < accept_ratio < 0.45                                       # metro.jl:15
    refcases.check_mean_std(thetas, case, case["tolm"])                     # metro.jl:16
    assert thetas.shape == ((case["niter"] // 2,) if np.ndim(case["theta0"]) == 0 else (case["niter"] // 2, 2))


def test_many_chains_recover_the_slow_reference_cases(kmc):
    """The reference needs niter = 10^7 on ONE chain for LogNormal and Rosenbrock (test/runtests.jl:57-61, :68-79);
    4096 chains x 20 000 steps give the same statistics (reference tolerances) from one launch."""
    cases = {c["name"]: c for c in refcases.CASES}
    c = cases["lognormal(0,1)"]
    th, acc, logd, _ = kmc.metropolis_chains(kmc.LogNormal(0.0, 1.0), kmc.GaussianStep(c["mstep"]), np.full(4096, c["theta0"]),
                                             niter=20000, nthin=10, seed=5)
    assert th.shape == (4096, 1000) and logd.shape == (4096, 1000)
    assert 0.15 < acc.mean() < 0.45
    flat, accm, l, _ = kmc.squash_walkers(th, acc, logd)
    assert flat.shape == (4096 * 1000,)
    refcases.check_mean_std(flat, c, c["tolm"])
    c = cases["rosenbrock2"]
    th, acc, logd, _ = kmc.metropolis_chains(kmc.Rosenbrock(), kmc.GaussianStep(c["mstep"]), np.zeros((4096, 2)),
                                             niter=20000, nthin=10, seed=6)
    assert 0.15 < acc.mean() < 0.45
    refcases.check_mean_std(th.reshape(-1, 2), c, c["tolm"])


@pytest.mark.parametrize("nc,nd,niter,nburn,nthin", [(1, 1, 9000, 3000, 1), (1, 2, 5000, 1000, 3), (63, 3, 700, 100, 2), (64, 1, 4100, 0, 1),
                                                     (65, 8, 300, 50, 1), (200, 5, 400, 399, 1), (1, 8, 2500, 2499, 1), (130, 2, 3, 1, 1)])
def test_few_chains_read_their_draws_from_a_table(kmc, oracle, monkeypatch, nc, nd, niter, nburn, nthin, kmc_debug):
    """One chain is the reference's own call (src/samplers.jl:59-77): partial waves, tiles of the table that end inside a run
    (an LDS tile is 32 KiB: 2048 steps of one 1-D chain, 3 steps of 64 8-D chains), tables shorter than the run (KMC_DEBUG=metro-table-steps),
    all stored samples in burn-in -- against the oracle, and equal to the in-kernel draws to the last bit."""
    from kissmcmc_jl_amd.metropolis import run_chains
    th = 0.3 * np.random.default_rng(nc + nd).standard_normal((nc, nd))
    step = np.linspace(0.4, 0.9, nd)
    pdf = kmc.Rosenbrock(1.0, 100.0, 20.0) if nd >= 2 else kmc.GaussianIso(-1.0, 2.0)
    did, params = (oracle.ROSENBROCK, [1.0, 100.0, 20.0]) if nd >= 2 else (oracle.GAUSSIAN_ISO, [-1.0, 2.0])
    ref = oracle.metropolis(did, params, th, step, niter, nburn, nthin, 31)
    got = {}
    for mode, steps in (("1", None), ("1", "257"), ("0", None)):
        kmc_debug.set("metro-table", mode)
        if steps:
            kmc_debug.set("metro-table-steps", steps)
        else:
            kmc_debug.unset("metro-table-steps")
        r = run_chains(pdf, kmc.GaussianStep(step), th, niter, nburn, nthin, 31, moments=True)
        np.testing.assert_array_equal(r["naccept"], ref["naccept"])
        np.testing.assert_allclose(r["chain"], ref["chain"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(r["chain_logp"], ref["chain_logp"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["final_pos"], ref["final_pos"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(r["chain_sum"], ref["chain_sum"], rtol=1e-9, atol=1e-9)
        got[(mode, steps)] = r
    for k in ("chain", "chain_logp", "final_pos", "final_logp", "naccept", "chain_sum", "chain_sumsq"):
        np.testing.assert_array_equal(got[("1", None)][k], got[("0", None)][k], err_msg=k)
        np.testing.assert_array_equal(got[("1", "257")][k], got[("0", None)][k], err_msg=k)


def test_body_density_few_chains_run_in_registers(kmc, oracle, monkeypatch, kmc_debug):
    """A function-body density (CDensity) with the table: chains in registers instead of in memory; equal to the chain-in-memory
    kernel and to the menu density it restates."""
    from kissmcmc_jl_amd.metropolis import run_chains
    body = "double s = 0; for (int i = 0; i < n; ++i) { double d = (x[i] - p[0]) * p[1]; s += d * d; } return -0.5 * s;"
    th = np.random.default_rng(2).standard_normal((70, 3))
    out = {}
    for mode in ("1", "0"):
        kmc_debug.set("metro-table", mode)
        out[mode] = run_chains(kmc.CDensity(body, params=[0.2, 1.0 / 1.5]), kmc.GaussianStep(0.8), th, 600, 200, 2, 9, moments=True)
    menu = run_chains(kmc.GaussianIso(0.2, 1.5), kmc.GaussianStep(0.8), th, 600, 200, 2, 9, moments=True)
    for k in ("naccept", "chain", "final_pos"):
        np.testing.assert_array_equal(out["1"][k], out["0"][k], err_msg=k)
    np.testing.assert_array_equal(out["1"]["naccept"], menu["naccept"])
    np.testing.assert_allclose(out["1"]["chain"], menu["chain"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["1"]["chain_sum"], out["0"]["chain_sum"], rtol=1e-12)


def test_user_density_runs_in_the_metropolis_kernel(kmc, oracle, draws):
    """Runtime-compiled log-density (hiprtc) in the many-chain kernel = the menu Gaussian, bit for bit."""
    from kissmcmc_jl_amd.metropolis import run_chains
    th = np.random.default_rng(1).standard_normal((500, 3))
    user = kmc.ExprDensity("-0.5 * ((x - p[0]) * p[1]) * ((x - p[0]) * p[1])", params=[0.2, 1.0 / 1.5])
    a = run_chains(user, kmc.GaussianStep(0.8), th, 60, 20, 1, 9)
    b = run_chains(kmc.GaussianIso(0.2, 1.5), kmc.GaussianStep(0.8), th, 60, 20, 1, 9)
    np.testing.assert_array_equal(a["naccept"], b["naccept"])
    np.testing.assert_allclose(a["chain"], b["chain"], rtol=1e-12, atol=1e-12)


def test_minus_inf_start_is_carried_like_the_reference(kmc):
    """The reference's metropolis does not require pdf(theta0) > -Inf (src/samplers.jl:70): the first finite proposal
    is accepted (p1 - (-Inf) = Inf > log u)."""
    th, acc, logd, _ = kmc.metropolis_chains(kmc.Exponential(), kmc.GaussianStep(2.0), np.full(64, -1.0), niter=400, nburnin=0, seed=2)
    assert np.all(th[:, -1] >= 0.0) and np.all(np.isfinite(logd[:, -1]))


# ---- host route: ANY closures for pdf / sample_ppdf (reference src/samplers.jl:59-61), blobs (:100-118) ------------
def test_closure_pdf_with_device_proposal_equals_the_oracle(kmc, oracle):
    """pdf = a Python closure computing the oracle's density (so the log-pdfs are the oracle's own), sample_ppdf =
    GaussianStep: proposals, accept decisions and counters are the in-kernel stream's -> equal to kmco_metropolis."""
    from kissmcmc_jl_amd.metropolis import run_chains
    nc, nd, niter, nburn, nthin, seed = 96, 3, 150, 40, 2, 17
    th = 0.3 * np.random.default_rng(2).standard_normal((nc, nd))
    pdf = kmc.HostLogPdf(lambda X: oracle.logpdf_batch(oracle.GAUSSIAN_ISO, [0.2, 1.5], X), vectorized=True)
    r = run_chains(pdf, kmc.GaussianStep(0.8), th, niter, nburn, nthin, seed, moments=True)
    ref = oracle.metropolis(oracle.GAUSSIAN_ISO, [0.2, 1.5], th, 0.8, niter, nburn, nthin, seed)
    np.testing.assert_array_equal(r["naccept"], ref["naccept"])
    np.testing.assert_allclose(r["chain"], ref["chain"], rtol=1e-11, atol=1e-11)
    np.testing.assert_array_equal(r["chain_logp"], oracle.logpdf_batch(oracle.GAUSSIAN_ISO, [0.2, 1.5], r["chain"].reshape(-1, nd)).reshape(r["chain_logp"].shape))
    np.testing.assert_allclose(r["chain_sum"], ref["chain_sum"], rtol=1e-10, atol=1e-10)
    # and the in-kernel fast path gives the same chain
    fast = run_chains(kmc.GaussianIso(0.2, 1.5), kmc.GaussianStep(0.8), th, niter, nburn, nthin, seed)
    np.testing.assert_array_equal(fast["naccept"], r["naccept"])
    np.testing.assert_allclose(fast["chain"], r["chain"], rtol=1e-11, atol=1e-11)


def test_closure_proposal_with_device_density_equals_a_host_restatement(kmc, oracle):
    """sample_ppdf = a deterministic Python closure, pdf = a menu density evaluated on the device: the chain must equal a
    line-by-line host loop of src/samplers.jl:96-126 fed the same proposals and the stream's accept uniforms."""
    from kissmcmc_jl_amd.metropolis import HostProposal, run_chains
    nc, nd, niter, nburn, nthin, seed = 8, 2, 60, 20, 3, 5
    th = np.linspace(-1, 1, nc * nd).reshape(nc, nd)
    calls = {"n": 0}

    def prop(X):                                  # symmetric in distribution is the user's business; here: a fixed schedule
        calls["n"] += 1
        return X + 0.4 * np.cos(calls["n"] + np.arange(X.size).reshape(X.shape))
    r = run_chains(kmc.GaussianIso(), HostProposal(prop, vectorized=True), th, niter, nburn, nthin, seed)
    x = th.copy()
    p0 = np.array([oracle.logpdf(oracle.GAUSSIAN_ISO, [0.0, 1.0], v) for v in x])
    nacc = np.zeros(nc, dtype=np.int64)
    chain = []
    for it in range(niter):
        n = it + 1 - nburn
        y = x + 0.4 * np.cos((it + 1) + np.arange(x.size).reshape(x.shape))
        for c in range(nc):
            p1 = oracle.logpdf(oracle.GAUSSIAN_ISO, [0.0, 1.0], y[c])
            ua = oracle.metropolis_draw(seed, it, c, 1)[1]
            if p1 - p0[c] > np.log(ua):                                   # :101
                x[c], p0[c] = y[c], p1
                nacc[c] += n > 0
        if n > 0 and n % nthin == 0:
            chain.append(x.copy())
    np.testing.assert_array_equal(r["naccept"], nacc)
    np.testing.assert_allclose(r["chain"], np.array(chain), rtol=1e-12, atol=1e-12)


def test_readme_style_call_with_two_closures_and_blobs(kmc):
    """metropolis(pdf, sample_ppdf, theta0) with two plain closures -- the reference's signature (src/samplers.jl:59) --
    and hasblob: the blob of every stored sample is the blob of the state that was stored (:100-103, :116-118)."""
    rng = np.random.default_rng(0)
    thetas, acc, logd, blobs = kmc.metropolis(lambda x: -0.5 * (x + 5.0) ** 2 / 9.0, lambda x: x + 1.5 * rng.standard_normal(), -5.0,
                                              niter=4000, use_progress_meter=False, seed=3)
    assert thetas.shape == (2000,) and logd.shape == (2000,) and blobs is None and 0.3 < acc < 0.95
    assert abs(thetas.mean() + 5.0) < 1.0 and abs(thetas.std() - 3.0) < 1.0
    np.testing.assert_allclose(logd, -0.5 * (thetas + 5.0) ** 2 / 9.0, rtol=1e-13)
    th, acc, logd, blobs = kmc.metropolis(lambda x: (-0.5 * float(x @ x), (x.copy(), "tag")), kmc.GaussianStep(0.7), np.array([0.1, -0.2]),
                                          niter=600, nburnin=100, nthin=5, hasblob=True, seed=9)
    assert th.shape == (100, 2) and len(blobs) == 100
    for k in range(100):
        np.testing.assert_array_equal(blobs[k][0], th[k])
        assert blobs[k][1] == "tag"
    # many chains, sum-reducing blobs (the reference's second blob case, test/runtests.jl:96-107)
    th, acc, logd, blobs = kmc.metropolis_chains(lambda x: (-0.5 * x * x, x), kmc.GaussianStep(1.0), np.zeros(16), niter=300, nburnin=100,
                                                 hasblob=True, init_blobs=lambda b0, ns: [0.0], reduce_blob=lambda bs, b: bs.__setitem__(0, bs[0] + b), seed=4)
    np.testing.assert_allclose([b[0] for b in blobs], th.sum(axis=1), rtol=1e-12, atol=1e-12)


def test_exceptions_in_host_closures_are_re_raised(kmc):
    def bad(x):
        raise KeyError("boom")
    with pytest.raises(KeyError):
        kmc.metropolis(kmc.GaussianIso(), bad, 0.0, niter=10)
