"""GPU: runtime-compiled user log-densities (ExprDensity, hiprtc) -- the device-side stand-in for the
reference's arbitrary `pdf` closure (src/samplers.jl:257).  Expressions that restate a menu density
must reproduce the oracle's chains for that density: identical accept decisions and positions,
log-pdfs to 1e-12 (different but equivalent arithmetic)."""
import numpy as np
import pytest

import kmcenv

pytestmark = pytest.mark.gpu

GAUSS = ("-0.5*x*x", None, [])
ROSEN = ("d < n-1 ? -((p[0]-x)*(p[0]-x))/p[2] : 0.0", "-(p[1]*((y-x*x)*(y-x*x)))/p[2]", [1.0, 100.0, 20.0])
EXPO = ("x < 0.0 ? -INFINITY : -p[0]*x", None, [1.0])


def _run(kmc, pdf, th, G, nburn, seed):
    nw, nd = th.shape
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        ch, cl = s.chain()
        m = s.moments()
        return dict(pos=s.positions(), logp=s.logp(), nacc=s.naccept(), chain=ch, chain_logp=cl, sum=m[0], n=m[2])


def _check(oracle, did, params, th, G, nburn, seed, got):
    nw, nd = th.shape
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, 1, 2.0, seed), th)
    np.testing.assert_array_equal(got["nacc"], ref["naccept"])
    np.testing.assert_array_equal(got["pos"], ref["final_pos"])
    np.testing.assert_array_equal(got["chain"], ref["chain"])
    tol = 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"]))
    assert np.all(np.abs(got["logp"] - ref["final_logp"]) <= tol)
    assert got["n"] == ref["nmoment"]
    np.testing.assert_allclose(got["sum"], ref["sum"], rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("nw,nd,plan", [(256, 32, ""), (200, 5, ""), (128, 1, ""), (256, 32, "generic"), (600, 200, "")])
def test_expr_gaussian_equals_menu_gaussian(kmc, oracle, nw, nd, plan, monkeypatch):
    if plan:
        monkeypatch.setenv("KMC_PLAN", plan)
    th = np.random.default_rng(nd).standard_normal((nw, nd))
    pdf = kmc.ExprDensity(GAUSS[0])
    got = _run(kmc, pdf, th, 80, 20, 7)          # one graph replay + eager tail
    _check(oracle, oracle.GAUSSIAN_ISO, [0.0, 1.0], th, 80, 20, 7, got)


@pytest.mark.parametrize("nw,nd", [(100, 2), (256, 64), (130, 9), (2100, 1024)])
def test_expr_rosenbrock_pair_term(kmc, oracle, nw, nd):
    th = 0.1 * np.random.default_rng(nd).standard_normal((nw, nd))
    pdf = kmc.ExprDensity(ROSEN[0], ROSEN[1], ROSEN[2])
    G = 60 if nd < 1000 else 4
    got = _run(kmc, pdf, th, G, G // 3, 11)
    _check(oracle, oracle.ROSENBROCK, ROSEN[2], th, G, G // 3, 11, got)


def test_expr_exponential_rejects_with_minus_infinity(kmc, oracle):
    th = 0.5 + 0.1 * np.abs(np.random.default_rng(1).standard_normal((100, 1)))
    pdf = kmc.ExprDensity(EXPO[0], params=EXPO[2])
    got = _run(kmc, pdf, th, 300, 150, 3)
    _check(oracle, oracle.EXPONENTIAL, [1.0], th, 300, 150, 3, got)
    assert got["pos"].min() >= 0.0


def test_expr_density_through_the_drop_in_api(kmc):
    """README sequence with a user expression instead of a menu density; make_theta0s evaluates the
    expression on the device for its `pdf(theta) > -Inf` test (src/samplers.jl:336-338)."""
    logpdf = kmc.ExprDensity("x < 0.0 ? -INFINITY : -x")        # README.md:15
    assert logpdf(0.5) == -0.5 and logpdf(-1.0) == -np.inf
    theta0s = kmc.make_theta0s(0.05, 0.1, logpdf, 100, rng=5)   # a third of the first tries are rejected
    assert theta0s.shape == (100,) and np.all(theta0s >= 0)
    thetas, acc, logd, blobs = kmc.emcee(logpdf, theta0s, niter=10 ** 5, use_progress_meter=False, seed=8)
    t, mean_acc = kmc.squash_walkers(thetas, acc)[:2]
    assert t.shape == (50000,) and abs(mean_acc - 0.745) < 0.02
    assert abs(t.mean() - 1.0) < 0.08 and t.min() >= 0.0
    np.testing.assert_allclose(logd, -thetas, rtol=0, atol=1e-15)


def test_expr_density_at_c2_size_speed_and_moments(kmc):
    """Full C2 shape with a user expression: same kernel structure, so the same ballpark speed."""
    th = np.random.default_rng(3).standard_normal((65536, 32))
    pdf = kmc.ExprDensity("-0.5*x*x")
    with kmc.Sampler(pdf, 65536, 32, 2000, 1000, 1, 2.0, 5, moments=True) as s:
        s.set_positions(th)
        s.run(2000)
        s.sync()
        ms = s.last_run_ms()
        msum, msq, n = s.moments()
    mean = msum / n
    var = msq / n - mean ** 2
    assert np.all(np.abs(mean) < 0.02) and np.all(np.abs(var - 1.0) < 0.02)
    assert ms / 4000 * 1e3 < 12.0, f"{ms / 4000 * 1e3:.2f} us per half-step"


# ---- body densities (CDensity, kmc_user_density_create_body): the whole function over the proposal vector --------------
C_GAUSS = "double s = 0.0; for (int i = 0; i < n; ++i) { double t = (x[i] - p[0]) * p[1]; s += t * t; } return -0.5 * s;"
C_ROSEN = ("double s = 0.0; for (int i = 0; i + 1 < n; ++i) { double d = x[i + 1] - x[i] * x[i]; double e = p[0] - x[i]; s += p[1] * (d * d) + e * e; } "
           "return -(s * (1.0 / p[2]));")
C_EXPO = "double s = 0.0; for (int i = 0; i < n; ++i) { if (x[i] < 0.0) return -INFINITY; s += x[i]; } return -(p[0] * s);"


@pytest.mark.parametrize("kernel", ["default", "one-walker-per-lane"])
@pytest.mark.parametrize("case", ["gauss_256x32", "gauss_1000x7", "rosen_128x2", "rosen_512x64", "expo_100x1", "expo_300x16",
                                  "gauss_200x33", "rosen_130x63", "gauss_96x65", "rosen_200x130", "gauss_256x32_generic",
                                  "gauss_300x256", "expo_170x97", "gauss_300x257", "rosen_196x130_generic"])
def test_body_density_equals_the_oracle(kmc, oracle, case, kernel, monkeypatch, kmc_debug):
    """A C++ function body that restates a menu density in the oracle's own element order: chains, counters AND log-pdfs
    equal the oracle's (the body is evaluated per walker, elements in index order, as the oracle does).  By default the rows
    travel lane-striped in the vector kernel and only the evaluation is per walker; with `no-body-vec` (and for blobs) the
    one-walker-per-lane kernels: staged through LDS up to 256 dimensions, generic above."""
    name, shape = case.split("_")[:2]
    kmc_debug.set("no-body-routing")                     # a general body is the subject: these ones are sums over elements and would
                                                         #  otherwise run lane-striped (test_separable_body_runs_lane_striped_and_equals_the_oracle)
    if kernel == "one-walker-per-lane":
        if case.endswith("_generic"):
            pytest.skip("runs one walker per lane anyway")
        kmc_debug.set("no-body-vec")
    kmcenv.no_resident(monkeypatch)           # the multi-launch kernels are the subject here (small ensembles with ndim <= 32
                                                         #  would run resident: test_body_density_runs_resident_on_small_ensembles)
    kmc_debug.set("fused", 0)                            #  ... and short rows one launch per generation: tests/test_gpu_generation.py
    if case.endswith("_generic"):
        monkeypatch.setenv("KMC_PLAN", "generic")        # the unstaged one-walker-per-lane kernel (what ndim > 256 runs)
    nw, nd = (int(v) for v in shape.split("x"))
    body, did, params, cparams, scale = {"gauss": (C_GAUSS, oracle.GAUSSIAN_ISO, [0.3, 1.5], [0.3, 1.0 / 1.5], 1.0),
                                         "rosen": (C_ROSEN, oracle.ROSENBROCK, [1.0, 100.0, 20.0], [1.0, 100.0, 20.0], 0.1),
                                         "expo": (C_EXPO, oracle.EXPONENTIAL, [1.0], [1.0], None)}[name]
    rng = np.random.default_rng(7)
    th = 0.5 + 0.1 * np.abs(rng.standard_normal((nw, nd))) if scale is None else scale * rng.standard_normal((nw, nd))
    pdf = kmc.CDensity(body, params=cparams)
    G, nburn, seed = 90, 25, 31
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed) as s:
        how = s.describe()
        want = ("half_step_generic" if case.endswith("_generic") else "half_step_vec" if kernel == "default" else
                "half_step_staged" if nd <= 256 else "half_step_generic")
        assert want in how and ("evaluated per walker" in how) == (want == "half_step_vec"), how
    got = _run(kmc, pdf, th, G, nburn, seed)
    _check(oracle, did, params, th, G, nburn, seed, got)
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, 1, 2.0, seed), th)
    if name != "gauss":                                  # same operations in the same order: not even a rounding difference
        np.testing.assert_array_equal(got["logp"], ref["final_logp"])
    assert pdf(th[0]) == pytest.approx(oracle.logpdf(did, params, th[0]), rel=1e-14)       # host call = device evaluation


@pytest.mark.parametrize("case", ["gauss_256x32", "gauss_1000x7", "rosen_128x2", "rosen_512x64", "gauss_200x33", "rosen_130x63", "gauss_96x65",
                                  "rosen_200x130", "gauss_300x256", "gauss_300x257", "gauss_2200x4", "rosen_4096x32"])
def test_separable_body_runs_lane_striped_and_equals_the_oracle(kmc, oracle, case, monkeypatch, kmc_debug):
    """A function body that is a sum over elements (`double s = 0; for (i < n) s += f(x[i]); return g(s);`, or the neighbour form with
    x[i + 1]) is recognised by kmc_user_density_create_body and runs in the lane-striped vector kernels of the menu densities: same
    chains and counters as the oracle, bit for bit; log-pdfs to rounding (lane-order sum).  The user's closure pdf(theta), src/samplers.jl:257."""
    name, shape = case.split("_")[:2]
    kmcenv.no_resident(monkeypatch)
    kmc_debug.set("fused", 0)                            # (short rows would run one launch per generation, the body as written: tests/test_gpu_generation.py)
    nw, nd = (int(v) for v in shape.split("x"))
    body, did, params, cparams, scale = {"gauss": (C_GAUSS, oracle.GAUSSIAN_ISO, [0.3, 1.5], [0.3, 1.0 / 1.5], 1.0),
                                         "rosen": (C_ROSEN, oracle.ROSENBROCK, [1.0, 100.0, 20.0], [1.0, 100.0, 20.0], 0.1)}[name]
    th = scale * np.random.default_rng(7).standard_normal((nw, nd))
    pdf = kmc.CDensity(body, params=cparams)
    assert pdf.separable
    G, nburn, seed = 90, 25, 31
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed) as s:
        how = s.describe()
        assert "half_step_vec" in how and "recognised as a sum over elements" in how
    got = _run(kmc, pdf, th, G, nburn, seed)
    _check(oracle, did, params, th, G, nburn, seed, got)
    assert pdf(th[0]) == pytest.approx(oracle.logpdf(did, params, th[0]), rel=1e-14)       # the host call still evaluates the body itself


@pytest.mark.parametrize("form", ["expr", "sum-body", "coupled-body"])
def test_offline_compiled_kernels_equal_the_oracle(kmc, oracle, form, monkeypatch, kmc_debug):
    """The same kernels built by hipcc as a child process (what the samplers choose by themselves for >= 16 384 walkers; forced here
    with KMC_DEBUG=rtc=hipcc) instead of hiprtc: chains, counters and moments equal to the oracle as ever."""
    kmc_debug.set("rtc", "hipcc")
    kmcenv.no_resident(monkeypatch)
    nw, nd, G, nburn, seed = 1536, 32, 90, 25, 31
    th = np.random.default_rng(7).standard_normal((nw, nd))
    if form == "expr":
        pdf, did, params = kmc.ExprDensity("-0.5*((x-p[0])*p[1])*((x-p[0])*p[1])", None, [0.3, 1.0 / 1.5]), oracle.GAUSSIAN_ISO, [0.3, 1.5]
    elif form == "sum-body":
        pdf, did, params = kmc.CDensity(C_GAUSS, params=[0.3, 1.0 / 1.5]), oracle.GAUSSIAN_ISO, [0.3, 1.5]
    else:
        kmc_debug.set("no-body-routing")
        pdf, did, params = kmc.CDensity(C_GAUSS, params=[0.3, 1.0 / 1.5]), oracle.GAUSSIAN_ISO, [0.3, 1.5]
    got = _run(kmc, pdf, th, G, nburn, seed)
    _check(oracle, did, params, th, G, nburn, seed, got)


def test_big_ensembles_ask_for_the_offline_compiler_by_themselves(kmc, oracle, tmp_path, monkeypatch):
    """>= 16 384 walkers in the multi-launch kernels: the vector kernel's code object comes from the offline compiler (a second cache entry
    next to hiprtc's), and the sampler is the oracle's."""
    monkeypatch.setenv("KMC_CACHE_DIR", str(tmp_path))
    pdf = kmc.CDensity(C_GAUSS, params=[0.3, 1.0 / 1.5])
    n0 = len(list(tmp_path.glob("*.co")))
    nw, nd, G, nburn, seed = 16384, 8, 40, 10, 5
    th = np.random.default_rng(3).standard_normal((nw, nd))
    got = _run(kmc, pdf, th, G, nburn, seed)
    assert len(list(tmp_path.glob("*.co"))) == n0 + 1
    _check(oracle, oracle.GAUSSIAN_ISO, [0.3, 1.5], th, G, nburn, seed, got)


@pytest.mark.parametrize("case", ["two-sums_3000x6", "two-sums_512x33", "three-sums-neighbour_1200x32", "three-sums-neighbour_300x130"])
def test_bodies_feeding_several_sums_run_lane_striped(kmc, case, monkeypatch, kmc_debug):
    """`double s = 0, t = 0; for (i < n) { s += f(x[i]); t += g(x[i]); } return h(s, t);` -- up to four sums fed by one pass over the
    elements (the neighbour form included) -- is recognised like the one-sum form and runs lane-striped (SepDensityN); the same
    body evaluated per walker (`no-body-routing`) gives the same chain and counters, log-pdfs to rounding (lane-order sums)."""
    name, shape = case.split("_")
    nw, nd = (int(v) for v in shape.split("x"))
    body = ("double s = 0.0, t = 0.0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += x[i]; } return -0.5 * (s + p[0] * t * t);" if name == "two-sums" else
            "double s = 0.0; double t = 0.0; double u = 0; for (int i = 0; i + 1 < n; ++i) { double d = x[i+1]-x[i]; s += d*d; t += x[i]*x[i]; u += x[i+1]*x[i]; } "
            "return -(p[0]*s + 0.5*t + 0.01*u);")
    kmcenv.no_resident(monkeypatch)
    kmc_debug.set("fused", 0)                            # (3000 x 6 would run one launch per generation, the body as written)
    th = np.random.default_rng(4).standard_normal((nw, nd))
    G, nburn, seed = 80, 20, 17
    routed = kmc.CDensity(body, params=[0.3])
    assert routed.separable
    with kmc.Sampler(routed, nw, nd, G, nburn, 1, 2.0, seed) as s:
        assert "recognised as a sum over elements" in s.describe()
    a = _run(kmc, routed, th, G, nburn, seed)
    kmc_debug.set("no-body-routing")
    plain = kmc.CDensity(body, params=[0.3])
    assert not plain.separable
    with kmc.Sampler(plain, nw, nd, G, nburn, 1, 2.0, seed) as s:
        assert "evaluated per walker" in s.describe()
    b = _run(kmc, plain, th, G, nburn, seed)
    for k in ("pos", "nacc", "chain"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert np.all(np.abs(a["logp"] - b["logp"]) <= 1e-12 * np.maximum(1.0, np.abs(b["logp"])))
    assert a["n"] == b["n"]
    np.testing.assert_allclose(a["sum"], b["sum"], rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("outcome", ["agree", "fail", "blind-forced", "blind"])
def test_sum_form_is_checked_against_the_body_before_it_runs(kmc, outcome, monkeypatch, kmc_debug, capfd):
    """The recogniser reads text.  Before a sampler runs the generated per-element form, that form and the body itself are evaluated
    on 256 test rows (kmc_sampler.hip: check_sum_form): agreement -> lane-striped; a difference (forced here: sum-form-check=fail) or
    no test row with a finite value (forced, and for real: a density supported on [100, 101] only) -> the density is evaluated per walker,
    as written, describe() says why, and the run equals the run with the recogniser switched off."""
    kmcenv.no_resident(monkeypatch)
    kmc_debug.set("fused", 0)                            # (the lane-striped route of the two-launch kernels is the subject)
    nw, nd, G, nburn, seed = 256, 8, 60, 10, 5
    if outcome == "blind":
        body = ("double s = 0; for (int i = 0; i < n; ++i) { const double t = x[i] - 100.5; s += (t < -0.5 || t > 0.5) ? -INFINITY : -0.5 * t * t; } return s;")
        th = 100.5 + 0.1 * np.random.default_rng(3).standard_normal((nw, nd))
    else:
        body = "double s = 0; for (int i = 0; i < n; ++i) { const double t = x[i] - p[0]; s += t * t; } return -0.5 * s;"
        th = np.random.default_rng(3).standard_normal((nw, nd))
    if outcome in ("fail", "blind-forced"):
        kmc_debug.set("sum-form-check", outcome.split("-")[0])
    pdf = kmc.CDensity(body, params=[0.3])
    assert pdf.separable                                   # the recogniser took it ...
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed) as s:
        how = s.describe()
    err = capfd.readouterr().err
    if outcome == "agree":
        assert "recognised as a sum over elements and checked against the body" in how and pdf.separable
        return
    # ... and the check took it back: for good where the two forms DISAGREE (the recogniser misread the body), for this sampler's (ndim, parameters)
    # only where no test row was finite (nothing shown there; another row length or parameter set is checked on its own -- ADVICE r04)
    assert pdf.separable == (outcome != "fail")
    assert "evaluated per walker" in how and "taken for a sum over elements, but" in how
    assert ("disagrees with the body" in how and "please report" in err) if outcome == "fail" else ("could not be checked" in how and "please report" not in err)
    a = _run(kmc, pdf, th, G, nburn, seed)
    kmc_debug.set("no-body-routing")
    plain = kmc.CDensity(body, params=[0.3])
    b = _run(kmc, plain, th, G, nburn, seed)
    for k in ("pos", "nacc", "chain", "logp"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["nacc"].sum() > 0


def test_sum_form_verdict_belongs_to_one_ndim_and_parameter_set(kmc, oracle, monkeypatch, kmc_debug):
    """A density whose support moves with a parameter: at p = 100.5 no test row is finite -- that sampler runs the body as written; at p = 0 (same
    density object created anew, and the SAME object with other parameters cannot exist: parameters belong to the density) and at another ndim the check
    sees finite rows and the routing stands.  A blind check no longer switches the routing off for every later sampler of the density."""
    kmcenv.no_resident(monkeypatch)
    kmc_debug.set("fused", 0)
    body = "double s = 0; for (int i = 0; i < n; ++i) { const double t = x[i] - p[0]; s += (t < -3.0 || t > 3.0) ? -INFINITY : -0.5 * t * t; } return s;"
    blind = kmc.CDensity(body, params=[100.5])
    with kmc.Sampler(blind, 256, 8, 20, 0, 1, 2.0, 1) as s:
        assert "could not be checked" in s.describe() and "evaluated per walker" in s.describe()
    assert blind.separable                                  # (not refuted: only unchecked there)
    with kmc.Sampler(blind, 256, 8, 20, 0, 1, 2.0, 1) as s:        # the same case again: decided already, the same way
        assert "could not be checked" in s.describe()
    seen = kmc.CDensity(body, params=[0.0])
    for nd in (8, 16):
        with kmc.Sampler(seen, 256, nd, 20, 0, 1, 2.0, 1) as s:
            assert "recognised as a sum over elements and checked against the body" in s.describe(), s.describe()
    # ... and the unrouted sampler is the run with the recogniser switched off
    th = 100.5 + 0.1 * np.random.default_rng(3).standard_normal((256, 8))
    a = _run(kmc, blind, th, 60, 10, 5)
    kmc_debug.set("no-body-routing")
    b = _run(kmc, kmc.CDensity(body, params=[100.5]), th, 60, 10, 5)
    for k in ("pos", "nacc", "chain", "logp"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["nacc"].sum() > 0


def test_body_density_with_real_coupling_samples_its_target(kmc, monkeypatch, kmc_debug):
    """A density no term / pair form can express -- a correlated Gaussian with a dense precision matrix built in the body:
    x' P x with P = (1 + rho) I - rho/n 11' ... here: -0.5 (sum x_i^2 + c (sum x_i)^2): variance of the mean direction
    1 / (1 + c n), of every orthogonal direction 1."""
    n, c = 6, 0.5
    kmcenv.no_resident(monkeypatch)           # (2048 x 6 would run resident, two walkers per thread: the multi-launch kernel is the subject)
    kmc_debug.set("fused", 0)                            # (... or one launch per generation)
    pdf = kmc.CDensity("double s = 0.0, t = 0.0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += x[i]; } return -0.5 * (s + p[0] * t * t);", params=[c])
    th = np.random.default_rng(1).standard_normal((2048, n))
    with kmc.Sampler(pdf, 2048, n, 3000, 500, 5, 2.0, 3, store_chain=True) as s:
        assert "half_step_vec" in s.describe() and "recognised as a sum over elements" in s.describe() and "runtime-compiled" in s.describe()
        s.set_positions(th)
        s.run(3000)
        s.sync()
        ch, _ = s.chain(logp=False)
    flat = ch.reshape(-1, n)
    mean_dir = flat.sum(axis=1) / np.sqrt(n)
    assert abs(mean_dir.var() - 1.0 / (1.0 + c * n)) < 0.02
    ortho = flat[:, 0] - flat[:, 1]
    assert abs(ortho.var() / 2.0 - 1.0) < 0.04


def test_body_density_other_entry_points(kmc, oracle):
    """The same handle in the device-side initial ball and in the many-chain Metropolis kernel; errors are reported."""
    from kissmcmc_jl_amd.metropolis import run_chains
    pdf = kmc.CDensity(C_EXPO, params=[1.0])
    ref = oracle.init_ball(oracle.EXPONENTIAL, [1.0], 0.02, 0.1, 2048, 3, seed=7)
    with kmc.Sampler(pdf, 2048, 3, 10) as s:
        s.init_ball(0.02, 0.1, seed=7)
        np.testing.assert_allclose(s.positions(), ref["pos"], rtol=1e-11, atol=1e-13)
    thm = 0.1 * np.random.default_rng(2).standard_normal((300, 2))
    a = run_chains(kmc.CDensity(C_ROSEN, params=[1.0, 100.0, 20.0]), kmc.GaussianStep(0.5), thm, 120, 40, 2, 11)
    b = oracle.metropolis(oracle.ROSENBROCK, [1.0, 100.0, 20.0], thm, 0.5, 120, 40, 2, 11)
    np.testing.assert_array_equal(a["naccept"], b["naccept"])
    np.testing.assert_allclose(a["chain"], b["chain"], rtol=1e-11, atol=1e-11)
    with pytest.raises(kmc.KmcError, match="does not compile"):
        kmc.CDensity("return x[0] +;")
    with pytest.raises(kmc.KmcError, match="KMC_ISLANDS"):
        kmc.Sampler(pdf, 1024, 3, 10, island_gens=8)


ROSEN_BODY = ("double s = 0.0; for (int i = 0; i + 1 < n; ++i) { double d = x[i + 1] - x[i] * x[i]; double e = p[0] - x[i]; s += p[1] * (d * d) + e * e; } "
              "return -(s * (1.0 / p[2]));")


@pytest.mark.parametrize("nw,nd,G,nburn,nthin", [(100, 2, 700, 300, 1), (6, 4, 300, 100, 3), (1000, 5, 150, 40, 2), (1024, 4, 120, 0, 1), (200, 31, 90, 30, 1),
                                                  (64, 1, 400, 100, 1), (1000, 8, 100, 30, 1), (512, 32, 60, 20, 1), (1024, 12, 60, 10, 2), (600, 24, 50, 10, 1),
                                                  (1500, 3, 90, 30, 2), (2048, 1, 70, 0, 1), (1030, 8, 60, 20, 1)])
def test_body_density_runs_resident_on_small_ensembles(kmc, oracle, monkeypatch, nw, nd, G, nburn, nthin, kmc_debug):
    """A CDensity on the reference's own problem sizes: the whole ensemble in one workgroup's LDS, one walker per thread, many
    generations per launch (kmc_islands.hpp: resident_lane_body) -- same draws and element order as the multi-launch kernels:
    identical to the oracle's run of the menu density AND to the same sampler with KMC_DEBUG=no-resident (chain, log-pdfs, counters,
    moments), across run() pieces and a restart."""
    kmc_debug.set("no-body-routing")        # (the comparison below is with the one-walker-per-lane multi-launch kernels: same order, same bits)
    if nd == 1:
        pdf, did, params = kmc.CDensity("return x[0] < 0.0 ? -INFINITY : -x[0];"), oracle.EXPONENTIAL, [1.0]
        th = 0.5 + 0.1 * np.abs(np.random.default_rng(5).standard_normal((nw, nd)))
    else:
        pdf, did, params = kmc.CDensity(ROSEN_BODY, params=[1.0, 100.0, 20.0]), oracle.ROSENBROCK, [1.0, 100.0, 20.0]
        th = 0.1 * np.random.default_rng(5).standard_normal((nw, nd))
    seed = 31
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, nthin, 2.0, seed), th)

    def run():
        with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
            for attempt in range(2):
                s.set_positions(th)
                for n in (1, G // 3, G - 1 - G // 3):
                    s.run(n)
                s.sync()
            chain, clogp = s.chain()
            return dict(how=s.describe(), pos=s.positions(), logp=s.logp(), nacc=s.naccept(), chain=chain, clogp=clogp, mom=s.moments(),
                        launches=s.launch_count)

    res = run()
    assert "resident mode" in res["how"] and ("two walkers per thread" if nw > 1024 else "one walker per thread") in res["how"], res["how"]
    assert res["launches"] <= 6                                           # three run() pieces (draw table + resident kernel each), not 2 G launches
    kmcenv.no_resident(monkeypatch)
    ml = run()
    assert "resident" not in ml["how"]
    for got in (res, ml):
        np.testing.assert_array_equal(got["nacc"], ref["naccept"])
        np.testing.assert_array_equal(got["pos"], ref["final_pos"])
        np.testing.assert_array_equal(got["chain"], ref["chain"])
        assert np.all(np.abs(got["clogp"] - ref["chain_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["chain_logp"])))
        assert got["mom"][2] == ref["nmoment"]
        np.testing.assert_allclose(got["mom"][0], ref["sum"], rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(got["mom"][1], ref["sumsq"], rtol=1e-11, atol=1e-9)
    np.testing.assert_array_equal(res["clogp"], ml["clogp"])               # the same function evaluated in the same order: same bits
    np.testing.assert_array_equal(res["logp"], ml["logp"])


def test_body_density_with_blobs_runs_resident(kmc):
    """... and carries its blobs there too (hasblob=true on the reference's own sizes: 100 walkers)."""
    body = "const double t = x[0] + 5.0; const double lp = -(t * t) / 18.0; blob[0] = x[0]; blob[1] = lp; return lp;"
    pdf = kmc.CDensity(body, nblob=2)
    th = -4.0 + 0.1 * np.random.default_rng(2).standard_normal((100, 1))
    with kmc.Sampler(pdf, 100, 1, 1000, 500, 1, 2.0, 4, store_chain=True, store_logp=True, store_blobs=True) as s:
        s.set_positions(th)
        s.run(1000)
        s.sync()
        assert "resident mode" in s.describe() and s.launch_count <= 2        # (the draw table's kernel + the resident one)
        chain, clogp = s.chain()
        blobs = s.blobs(by_walker=False)
        cur, pos, lp = s.current_blobs(), s.positions(), s.logp()
    np.testing.assert_array_equal(blobs[:, :, 0], chain[:, :, 0])
    np.testing.assert_array_equal(blobs[:, :, 1], clogp)
    np.testing.assert_array_equal(cur[:, 0], pos[:, 0])
    np.testing.assert_array_equal(cur[:, 1], lp)
    assert abs(chain.mean() + 5.0) < 0.9


def test_term_pair_density_runs_two_walkers_per_thread_beyond_1024(kmc, oracle, monkeypatch):
    """An ExprDensity on 1026 .. 2048 walkers with short rows: resident mode with two walkers per thread, equal to the menu density's
    multi-launch run bit for bit."""
    nw, nd, G, nburn = 1800, 4, 120, 30
    th = np.random.default_rng(3).standard_normal((nw, nd))

    def run(pdf):
        with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, 17, store_chain=True, store_logp=True, moments=True) as s:
            s.set_positions(th)
            s.run(G)
            s.sync()
            return dict(how=s.describe(), chain=s.chain()[0], nacc=s.naccept(), mom=s.moments())

    a = run(kmc.ExprDensity("-0.5*x*x"))
    assert "two walkers per thread" in a["how"], a["how"]
    kmcenv.no_resident(monkeypatch)
    b = run(kmc.GaussianIso(0.0, 1.0))
    assert "resident" not in b["how"]
    np.testing.assert_array_equal(a["chain"], b["chain"])
    np.testing.assert_array_equal(a["nacc"], b["nacc"])
    np.testing.assert_allclose(a["mom"][0], b["mom"][0], rtol=1e-11, atol=1e-8)
