"""One launch per generation (kissmcmc.jl_amd/csrc/kmc_generation.hpp) against the CPU oracle and against the two-launch kernels.

Mid-size ensembles with short double rows run a whole generation (reference src/samplers.jl:246-274, both batches) per launch:
second-half walkers recompute their partner's first-half move instead of waiting for it.  The bars are those of
tests/test_gpu_parity.py: accept counters, positions and the chain bit-identical to the oracle, log-pdfs to 1e-12, moments to 1e-11.
"""
import numpy as np
import pytest

from test_gpu_parity import _compare, _densities, _run_both, _theta0

pytestmark = pytest.mark.gpu


def _mode(kmc, pdf, nw, nd, **kw):
    with kmc.Sampler(pdf, nw, nd, 10, 0, 1, 2.0, 1, **kw) as s:
        return s.describe()


@pytest.mark.parametrize("name,nw,nd", [
    ("gauss", 2050, 4), ("gauss", 4096, 4), ("gauss", 10000, 4), ("gauss", 16384, 4), ("gauss", 4096, 1), ("gauss", 4096, 2),
    ("gauss", 4098, 3), ("gauss_shift", 2500, 5), ("gauss", 4096, 7), ("gauss", 4096, 8), ("gauss", 65536, 2),
    ("expo", 4096, 1), ("expo", 3000, 3), ("rosen", 4096, 2), ("rosen", 2600, 6), ("lognormal", 4096, 1), ("lognormal", 5000, 4),
    # longer rows, lane-striped (generation_group): every row geometry of the vector kernels, ragged and odd row lengths
    ("gauss", 2048, 32), ("gauss", 4096, 32), ("rosen", 4096, 64), ("gauss", 1200, 100), ("gauss", 3000, 33), ("expo", 2600, 16),
    ("lognormal", 4096, 12), ("gauss", 512, 128), ("gauss", 600, 300), ("gauss", 520, 500), ("rosen", 1400, 9), ("gauss", 16384, 10),
    ("rosen", 2050, 130), ("gauss", 2000, 257),
])
def test_one_launch_per_generation_equals_the_oracle(kmc, oracle, kmc_debug, name, nw, nd):
    """The whole surface of a run: 70 generations = one graph chunk of 64 + 6 launched one by one, burn-in 13, every 3rd generation
    stored (chain, log-pdfs, moments), counters restarted at the end of burn-in.  (KMC_DEBUG=fused=1: the kernel wherever it exists,
    whatever the planner's size rule says -- test_what_keeps_the_two_launch_kernels is about that rule.)"""
    kmc_debug.set("fused", 1)
    pdf = _densities(kmc, oracle)[name][0]
    assert "one launch per generation" in _mode(kmc, pdf, nw, nd)
    ref, got = _run_both(kmc, oracle, name, nw, nd, 70, 13, 3, 11)
    _compare(ref, got)


@pytest.mark.parametrize("name,nw,nd", [("gauss", 4096, 7), ("gauss", 4096, 8), ("gauss_shift", 2500, 5), ("rosen", 2600, 6), ("expo", 3000, 5)])
def test_rows_of_five_to_eight_doubles_one_walker_per_lane(kmc, oracle, kmc_debug, name, nw, nd):
    """Rows of 5 ... 8 doubles run striped over a quad by default (generation_group<4, 1>: the cases of the test above); KMC_DEBUG=fused=lane
    keeps them one walker per lane -- the form bodies with real coupling take at these row lengths."""
    kmc_debug.set("fused", "lane")
    pdf = _densities(kmc, oracle)[name][0]
    assert "generation_lane" in _mode(kmc, pdf, nw, nd)
    ref, got = _run_both(kmc, oracle, name, nw, nd, 70, 13, 3, 11)
    _compare(ref, got)


@pytest.mark.parametrize("G,nburn,nthin", [(1, 0, 1), (2, 1, 1), (63, 0, 1), (64, 64, 1), (65, 10, 7), (129, 0, 2), (200, 199, 1)])
def test_run_lengths_and_schedules(kmc, oracle, G, nburn, nthin):
    """Odd and even numbers of generations (the state ends in either copy and is moved back), whole chunks and tails, burn-in up to
    the whole run, thinning."""
    ref, got = _run_both(kmc, oracle, "gauss", 4096, 4, G, nburn, nthin, 5)
    _compare(ref, got)


def test_split_runs_and_state_between_them(kmc, oracle):
    """kmc_sampler_run in pieces of odd lengths: between the calls the state is in the sampler's canonical arrays (read out, compared),
    and the pieces add up to the oracle's uninterrupted run."""
    nw, nd, G, nburn, seed = 4096, 4, 150, 20, 8
    th = _theta0("gauss", nw, nd, 3)
    pdf = kmc.GaussianIso()
    with kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
        assert "one launch per generation" in s.describe()
        s.set_positions(th)
        done = 0
        for piece in (1, 64, 3, 65, 17):
            s.run(piece)
            done += piece
            s.sync()
            part = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, done, min(nburn, done), 1, 2.0, seed), th, store_chain=False)
            np.testing.assert_array_equal(s.positions(), part["final_pos"])
        assert done == G
        got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio())
        got["chain"], got["chain_logp"] = s.chain()
        got["sum"], got["sumsq"], got["nmoment"] = s.moments()
        assert s.launch_count == G
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 1, 2.0, seed), th)
    _compare(ref, got)


def test_same_run_as_the_two_launch_kernels(kmc, kmc_debug):
    """KMC_DEBUG=fused=0 keeps the two-launch kernels for the same ensemble: positions, counters and chain bit for bit, log-pdfs to
    rounding (their sums run lane-striped)."""
    nw, nd, G, seed = 8192, 4, 100, 21
    th = _theta0("gauss", nw, nd, 9)
    out = {}
    for label in ("fused", "two"):
        if label == "two":
            kmc_debug.set("fused", 0)
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 10, 1, 2.0, seed, store_chain=True, moments=True) as s:
            assert ("one launch per generation" in s.describe()) == (label == "fused"), s.describe()
            s.set_positions(th)
            s.run(G)
            s.sync()
            out[label] = (s.positions(), s.naccept(), s.chain(logp=False)[0], s.logp(), s.moments())
    np.testing.assert_array_equal(out["fused"][0], out["two"][0])
    np.testing.assert_array_equal(out["fused"][1], out["two"][1])
    np.testing.assert_array_equal(out["fused"][2], out["two"][2])
    np.testing.assert_allclose(out["fused"][3], out["two"][3], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["fused"][4][0], out["two"][4][0], rtol=1e-11, atol=1e-9)
    assert out["fused"][4][2] == out["two"][4][2]


def test_what_keeps_the_two_launch_kernels(kmc):
    """Stepping by halves needs the half-step kernels (KMC_NO_GRAPH), and long rows / big ensembles stay with them."""
    pdf = kmc.GaussianIso()
    assert "one launch per generation" not in _mode(kmc, pdf, 4096, 4, use_graph=False)
    assert "one launch per generation" not in _mode(kmc, pdf, 65536, 32)
    assert "one launch per generation" not in _mode(kmc, pdf, 262144, 4)
    assert "one launch per generation" not in _mode(kmc, pdf, 65536, 16) and "one launch per generation" not in _mode(kmc, pdf, 20480, 64)       # 8 MiB beyond 49 152 walkers, 10 MiB
    assert "generation_group" in _mode(kmc, kmc.Rosenbrock(), 16384, 64) and "generation_group" in _mode(kmc, pdf, 32768, 32)   # 8 MiB of state, <= 49 152 walkers (C3; round 5, after the row masks went)
    assert "generation_lane" in _mode(kmc, pdf, 32768, 4) and "generation_group" in _mode(kmc, pdf, 8192, 32)
    assert "generation_group" in _mode(kmc, pdf, 16384, 32) and "generation_group" in _mode(kmc, pdf, 8192, 64)      # 4 MiB of state (round 5)
    assert "generation_group L=4 K=1" in _mode(kmc, pdf, 4096, 8) and "generation_group L=4 K=1" in _mode(kmc, pdf, 8192, 5)
    assert "generation_lane" in _mode(kmc, pdf, 16384, 6) and "generation_lane" in _mode(kmc, pdf, 4096, 4)
    assert "resident" in _mode(kmc, pdf, 2048, 4)
    # the short-row limit the documents quote (README, DESIGN section 4, scripts/feature_matrix.py: "ndim <= 8 up to 49 152 walkers", 196 608 doubles of state;
    # 65 536 walkers of one or two doubles): pinned here so that documents and planner cannot drift apart
    assert "generation_lane" in _mode(kmc, pdf, 49152, 4) and "one launch per generation" not in _mode(kmc, pdf, 65536, 4)
    assert "generation_group L=4 K=1" in _mode(kmc, pdf, 49152, 8) and "one launch per generation" not in _mode(kmc, pdf, 65536, 8)
    assert "generation_lane" in _mode(kmc, pdf, 65536, 2) and "one launch per generation" not in _mode(kmc, pdf, 65536, 3)


def test_stepping_by_halves_takes_the_two_launch_kernels_in_place(kmc, oracle):
    """kmc_sampler_half_step on a sampler that runs one launch per generation: it goes back to its two-launch kernels where it stands -- after
    generations already run, moments already credited -- and the whole run is still the oracle's (this used to be refused)."""
    for nw, nd in ((4096, 4), (4096, 32)):                  # generation_lane (per-sample sums) and generation_group (sojourn-weighted sums)
        G, nburn, seed = 90, 7, 5
        th = _theta0("gauss", nw, nd, 2)
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 2, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
            assert "one launch per generation" in s.describe()
            s.set_positions(th)
            s.run(41)
            for _ in range(9):
                s.half_step(0)
                s.half_step(1)
            assert "one launch per generation" not in s.describe()
            s.run(G - 50)
            s.sync()
            got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio())
            got["chain"], got["chain_logp"] = s.chain()
            got["sum"], got["sumsq"], got["nmoment"] = s.moments()
        ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 2, 2.0, seed), th)
        _compare(ref, got)


def test_streamed_chain_and_checkpoint(kmc, oracle, kmc_debug):
    """The chain streamed to host memory through the device ring (small blocks forced), and a run resumed from a checkpoint: both
    the oracle's uninterrupted run."""
    nw, nd, G, nburn, seed = 4096, 4, 300, 10, 13
    th = _theta0("gauss", nw, nd, 4)
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 1, 2.0, seed), th)
    kmc_debug.set("chain-block", 70)
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, seed, store_chain=True, store_logp=True, stream_chain=True) as s:
        assert "one launch per generation" in s.describe() and "streamed" in s.describe()
        s.set_positions(th)
        s.run(G)
        s.sync()
        chain, clogp = s.chain()
    kmc_debug.unset("chain-block")
    np.testing.assert_array_equal(chain, ref["chain"])
    np.testing.assert_allclose(clogp, ref["chain_logp"], rtol=1e-12, atol=1e-12)
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, seed, moments=True) as s:
        s.set_positions(th)
        s.run(77)
        s.sync()
        state = s.state()
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, seed, moments=True) as s:
        s.restore(state)
        s.run(G - 77)
        s.sync()
        np.testing.assert_array_equal(s.positions(), ref["final_pos"])
        np.testing.assert_array_equal(s.naccept(), ref["naccept"])


GAUSS_BODY = "double s = 0.0; for (int i = 0; i < n; ++i) { double t = (x[i] - p[0]) * p[1]; s += t * t; } return -0.5 * s;"
COUPLED_BODY = "double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; for (int i = 0; i + 1 < n; ++i) s += p[0] * x[i] * x[i + 1]; return -0.5 * s;"


@pytest.mark.parametrize("form", ["expr", "body", "two_sums"])
def test_runtime_compiled_densities_with_longer_rows(kmc, oracle, kmc_debug, form):
    """Lane-striped runtime-compiled densities in the one-launch-per-generation kernel (ndim 32): a term expression, a body recognised as
    a sum over elements, a body feeding two sums -- against the oracle where the Gaussian is restated, else against the same body in the
    two-launch kernels; a body with real coupling keeps the two-launch kernels."""
    nw, nd, G, nburn, nthin, seed = 4096, 32, 70, 13, 3, 11
    th = _theta0("gauss", nw, nd, seed)
    pdf = (kmc.ExprDensity("-0.5*x*x") if form == "expr" else kmc.CDensity(GAUSS_BODY, params=[0.0, 1.0]) if form == "body" else
           kmc.CDensity("double s = 0.0, t = 0.0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += x[i]; } return -0.5 * (s + p[0] * t * t);", params=[0.05]))

    def run():
        with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
            how = s.describe()
            s.set_positions(th)
            s.run(G)
            s.sync()
            got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio())
            got["chain"], got["chain_logp"] = s.chain()
            got["sum"], got["sumsq"], got["nmoment"] = s.moments()
            return how, got

    how, got = run()
    assert "one launch per generation" in how and "generation_group" in how and "runtime-compiled" in how, how
    if form == "two_sums":
        kmc_debug.set("fused", 0)
        how2, two = run()
        assert "half_step_vec" in how2
        ref = dict(two, status=0)
    else:
        ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, nthin, 2.0, seed), th)
    _compare(ref, got)
    with kmc.Sampler(kmc.CDensity(COUPLED_BODY, params=[0.3]), nw, nd, G, nburn, nthin, 2.0, seed) as s:
        assert "one launch per generation" not in s.describe()


@pytest.mark.parametrize("form", ["expr", "body", "body_as_written"])
def test_runtime_compiled_densities_equal_the_oracle(kmc, oracle, kmc_debug, form):
    """The caller's own density (reference src/samplers.jl:257) in the one-launch-per-generation kernel: a term expression, a C function
    body (a sum over elements: here it is evaluated as written, one walker per lane -- the oracle's element order), and the same body
    with the recogniser off.  The Gaussian restated: the oracle's run, bit for bit."""
    nw, nd, G, nburn, nthin, seed = 4096, 4, 70, 13, 3, 11
    if form == "body_as_written":
        kmc_debug.set("no-body-routing")
    pdf = kmc.ExprDensity("-0.5*x*x") if form == "expr" else kmc.CDensity(GAUSS_BODY, params=[0.0, 1.0])
    th = _theta0("gauss", nw, nd, seed)
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, nthin, 2.0, seed), th)
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
        how = s.describe()
        assert "one launch per generation" in how and "runtime-compiled" in how, how
        s.set_positions(th)
        s.run(G)
        s.sync()
        got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio())
        got["chain"], got["chain_logp"] = s.chain()
        got["sum"], got["sumsq"], got["nmoment"] = s.moments()
    _compare(ref, got)


def test_coupled_body_equals_its_two_launch_run(kmc, kmc_debug):
    """A body with a coupling between neighbouring elements (no menu density restates it): the one-launch-per-generation run against the
    same body in the two-launch kernels (KMC_DEBUG=fused=0)."""
    nw, nd, G, seed = 6000, 6, 90, 5
    th = _theta0("gauss", nw, nd, 2)
    out = {}
    for label in ("fused", "two"):
        if label == "two":
            kmc_debug.set("fused", 0)
        with kmc.Sampler(kmc.CDensity(COUPLED_BODY, params=[0.3]), nw, nd, G, 20, 2, 2.0, seed, store_chain=True, moments=True) as s:
            assert ("one launch per generation" in s.describe()) == (label == "fused"), s.describe()
            s.set_positions(th)
            s.run(G)
            s.sync()
            out[label] = (s.positions(), s.naccept(), s.chain(logp=False)[0], s.logp(), s.moments())
    np.testing.assert_array_equal(out["fused"][0], out["two"][0])
    np.testing.assert_array_equal(out["fused"][1], out["two"][1])
    np.testing.assert_array_equal(out["fused"][2], out["two"][2])
    np.testing.assert_allclose(out["fused"][3], out["two"][3], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["fused"][4][0], out["two"][4][0], rtol=1e-11, atol=1e-9)


def test_drop_in_call_on_a_mid_size_ensemble(kmc, oracle):
    """The reference's own call (src/samplers.jl:188) on 4 096 walkers: chain, accept ratios and log-densities as the oracle returns them."""
    nw, nd, gens, seed = 4096, 2, 40, 19
    th = _theta0("gauss", nw, nd, 6)
    thetas, accept_ratio, logdensities, blobs = kmc.emcee(kmc.GaussianIso(), th, niter=gens * nw, nburnin=10 * nw, use_progress_meter=False, seed=seed)
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, gens, 10, 1, 2.0, seed), th)
    assert blobs is None and thetas.shape == (nw, gens - 10, nd)
    np.testing.assert_array_equal(np.transpose(thetas, (1, 0, 2)), ref["chain"])
    np.testing.assert_array_equal(accept_ratio, ref["accept_ratio"])
    np.testing.assert_allclose(np.transpose(logdensities), ref["chain_logp"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("launch", [None, "updated"])
@pytest.mark.parametrize("nw,nd", [(4096, 4), (16384, 32), (65536, 16)])
def test_bound_position_buffer_and_callers_stream(kmc, oracle, monkeypatch, nw, nd, launch):
    """kmc_sampler_bind_positions + kmc_sampler_set_stream on a one-launch-per-generation sampler: the caller's buffer is the canonical
    copy of the state (current after every kmc_sampler_run, odd run lengths included), the second copy stays the library's.  The stream
    is torch's current one -- the legacy default stream, which cannot be captured: the graph chunk is recorded on a stream of the library's
    own and replayed on the caller's (the lane-striped form: 16384 x 32; the two-launch kernels likewise: 65536 x 16).  KMC_LAUNCH=updated: the same through the
    graph whose node parameters are rewritten before every replay (the bound buffer's address is among them)."""
    import torch
    if launch:
        monkeypatch.setenv("KMC_LAUNCH", launch)
    else:
        monkeypatch.delenv("KMC_LAUNCH", raising=False)
    G, seed = 131, 17
    th = _theta0("gauss", nw, nd, 8)
    pos = torch.zeros((nw, nd), dtype=torch.float64, device="cuda")
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 0, 1, 2.0, seed) as s:
        s.bind_positions(pos.data_ptr())
        s.set_stream(torch.cuda.current_stream().cuda_stream)
        assert ("one launch per generation" in s.describe()) == (nw <= 16384)
        s.set_positions(th)
        s.run(65)
        s.run(66)
        torch.cuda.synchronize()
        nacc = s.naccept()
        mine = pos.cpu().numpy()
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, 0, 1, 2.0, seed), th, store_chain=False)
    np.testing.assert_array_equal(mine, ref["final_pos"])
    np.testing.assert_array_equal(nacc, ref["naccept"])


@pytest.mark.parametrize("name,nw,nd", [("gauss", 4096, 4), ("gauss", 4096, 32), ("rosen", 4096, 64), ("expo", 2600, 16), ("gauss", 1200, 100),
                                        ("gauss", 3000, 33), ("lognormal", 5000, 4), ("rosen", 2600, 6)])
def test_two_launch_kernels_on_the_shapes_that_default_to_one(kmc, oracle, kmc_debug, name, nw, nd):
    """The two-launch kernels stay the path of every ensemble beyond the planner's size rule (and of every sharded one): on the small
    states that now run one launch per generation by default they are kept under test with KMC_DEBUG=fused=0 -- the oracle's run as well."""
    kmc_debug.set("fused", 0)
    pdf = _densities(kmc, oracle)[name][0]
    assert "one launch per generation" not in _mode(kmc, pdf, nw, nd)
    ref, got = _run_both(kmc, oracle, name, nw, nd, 70, 13, 3, 11)
    _compare(ref, got)


@pytest.mark.parametrize("nw,nd", [(4096, 4), (8192, 7), (4096, 32)])
def test_long_run_equals_the_two_launch_kernels(kmc, kmc_debug, nw, nd):
    """50 000 generations (780 graph replays + a tail) in each of the three forms -- one walker per lane, rows over a quad, rows lane-striped --
    against the same run on the two-launch kernels: positions and acceptance counters (no-return atomic adds in the new kernels) bit for bit,
    moments to rounding."""
    G, seed = 50007, 77
    th = _theta0("gauss", nw, nd, 12)
    out = {}
    for label in ("one", "two"):
        if label == "two":
            kmc_debug.set("fused", 0)
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 1000, 1, 2.0, seed, moments=True) as s:
            assert ("one launch per generation" in s.describe()) == (label == "one")
            s.set_positions(th)
            s.run(G)
            s.sync()
            out[label] = (s.positions(), s.naccept(), s.moments())
    np.testing.assert_array_equal(out["one"][0], out["two"][0])
    np.testing.assert_array_equal(out["one"][1], out["two"][1])
    assert out["one"][2][2] == out["two"][2][2]
    np.testing.assert_allclose(out["one"][2][0], out["two"][2][0], rtol=1e-10, atol=1e-6)
    np.testing.assert_allclose(out["one"][2][1], out["two"][2][1], rtol=1e-10)


@pytest.mark.parametrize("launch", ["graph", "updated", "eager", None])
@pytest.mark.parametrize("name,nw,nd", [("gauss", 4096, 4), ("rosen", 4096, 64), ("gauss", 3000, 33)])
def test_every_launch_mode_of_the_generation_kernels_is_the_oracles_run(kmc, oracle, monkeypatch, launch, name, nw, nd):
    """One launch per generation from the table graph, from a graph whose node parameters are rewritten before every replay (the generation among the
    preloaded kernel parameters: round 5), eagerly, or as measured (None: a run long enough to calibrate): uneven pieces -- whole chunks, odd tails, a
    piece that starts in the second copy of the state -- end in the oracle's uninterrupted run, chain and moments included."""
    if launch is None:
        monkeypatch.delenv("KMC_LAUNCH", raising=False)
    else:
        monkeypatch.setenv("KMC_LAUNCH", launch)
    pdf, did, params = _densities(kmc, oracle)[name]
    G, nburn, nthin, seed = 1200, 301, 7, 17
    th = _theta0(name, nw, nd, seed)
    planned = 4200 if launch is None else G                    # (nothing forced: the modes are measured for a job PLANNED long -- >= 4096 generations -- at its first call of >= 896; run here: 1200 of them)
    with kmc.Sampler(pdf, nw, nd, planned, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, moments=True) as s:
        assert "one launch per generation" in s.describe()
        s.set_positions(th)
        for piece in (1, 64, 65, 129, 1200 - 259):           # (the last piece is long enough -- 896 generations -- for the launch-mode measurement when nothing is forced)
            s.run(piece)
        s.sync()
        how = s.describe()
        got = dict(final_pos=s.positions(), final_logp=s.logp(), naccept=s.naccept(), accept_ratio=s.accept_ratio())
        got["chain"], got["chain_logp"] = s.chain()
        got["sum"], got["sumsq"], got["nmoment"] = s.moments()
        mode, _ = s.launch_mode()
        assert s.generation == G and s.launch_count == G
    if launch == "updated":
        assert mode == 3 and "generation preloaded" in how and "replay of 128 generations" in how, how
    elif launch == "graph":
        assert mode == 1 and "preloaded" not in how, how
    elif launch == "eager":
        assert mode == 2 and "eager launches" in how, how
    else:
        assert mode in (1, 2, 3) and "measured per 64 generations" in how, how
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, nthin, 2.0, seed), th)
    _compare(ref, got)


def test_generation_kernels_leave_the_updated_graph_when_the_budget_is_spent_or_the_caller_steps_by_halves(kmc, oracle, monkeypatch):
    """The updated-graph mode of a one-launch-per-generation sampler draws on the same process-wide budget of parameter updates (one per generation); it
    goes on with the table graph when that is spent in the middle of a run, and kmc_sampler_half_step takes the sampler to its two-launch kernels with
    the updated graph of generation kernels gone -- all of it the oracle's run."""
    import ctypes as C
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    used, budget = C.c_int64(0), C.c_int64(0)
    L.kmc_updated_budget(C.byref(used), C.byref(budget))
    L.kmc_set_updated_budget_mb(1.0)
    one_mib = C.c_int64(0)
    L.kmc_updated_budget(None, C.byref(one_mib))
    each = float(round(1048576.0 / one_mib.value))
    nw, nd, G, nburn, seed = 4096, 4, 700, 100, 23
    th = _theta0("gauss", nw, nd, seed)
    monkeypatch.setenv("KMC_LAUNCH", "updated,budget")
    try:
        L.kmc_set_updated_budget_mb((used.value + 3 * 128) * each / 1048576.0 + 1e-9)     # room for three replays of 128 generations, one update each (the check before a replay asks for 2 x 128: two get through)
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, seed, moments=True) as s:
            s.set_positions(th)
            s.run(640)
            s.sync()
            how = s.describe()
            mode, fell_back = s.launch_mode()
            assert fell_back and mode in (0, 1, 2) and "budget of the process spent" in how, (mode, fell_back, how)      # (0: too little left to measure -- whole chunks from the table graph)
            u2 = C.c_int64(0)
            L.kmc_updated_budget(C.byref(u2), None)
            assert 128 <= u2.value - used.value <= 3 * 128
            s.half_step(0)
            s.half_step(1)                                   # generation 641, by halves: the two-launch kernels from here on
            assert "one launch per generation" not in s.describe()
            s.run(G - 641)
            s.sync()
            pos, nacc = s.positions(), s.naccept()
            msum, msq, n = s.moments()
    finally:
        L.kmc_set_updated_budget_mb(budget.value * each / 1048576.0 + 1e-9)
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 1, 2.0, seed), th, store_chain=False)
    np.testing.assert_array_equal(pos, ref["final_pos"])
    np.testing.assert_array_equal(nacc, ref["naccept"])
    assert n == ref["nmoment"]
    np.testing.assert_allclose(msum, ref["sum"], rtol=1e-11, atol=1e-9)
