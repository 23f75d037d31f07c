"""GPU: the margin behind "identical accept decisions" (reference src/samplers.jl:260).

The device takes the two logarithms of the accept test -- t1 = (N-1) log z and lu = log u -- from its own routine
(kmc_device.hpp: log_pos_normal, the fdlibm algorithm, < 1 ulp), the oracle from glibc's log (< 1 ulp, a different function).
Everything else on the left of `>=` is bit-identical on both sides (positions are bit-identical, so are p0; p1 differs by
summation order, bounded separately).  Here: 10^7 seeded draws of the stream the samplers use, the device's values next to
the oracle's, the largest gap in ulps, and what that gap means: an accept decision can only differ where
|lhs - lu| <= gap(t1) + gap(lu), an interval the continuous variable log u hits with probability ~ gap * density(lu).
DESIGN.md section 6 quotes the numbers this test prints."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 10_000_000
SEED, STEP, NHALF, A, NDIM = 12345, 4242, 32768, 2.0, 32      # C2's stream: 65 536 walkers x 32 dims, a = 2


def ulp_gap(x, y):
    """|x - y| in units of the spacing of doubles at y (y != 0)."""
    return np.abs(x - y) / np.spacing(np.abs(y))


def test_accept_term_gap_and_flip_probability(kmc, oracle, capsys):
    import ctypes as C
    from kissmcmc_jl_amd import _lib
    z, t1, lu = np.empty(N), np.empty(N), np.empty(N)
    part = np.empty(N, dtype=np.int64)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    _lib.check(_lib.lib().kmc_debug_accept_terms(SEED, STEP, 0, N, NHALF, A, NDIM, 0, part.ctypes.data_as(ip), z.ctypes.data_as(dp),
                                                 t1.ctypes.data_as(dp), lu.ctypes.data_as(dp)))
    opart, oz, ot1, olu = oracle.accept_terms(SEED, STEP, 0, N, NHALF, A, NDIM)
    # the draws themselves are the same bits on both sides
    np.testing.assert_array_equal(part, opart)
    np.testing.assert_array_equal(z, oz)
    assert z.min() >= 1.0 / A and z.max() <= A
    # the logarithms: each side < 1 ulp of the true value, so at most 2 ulp apart (t1: of log z, then one rounding of the product)
    nz = oz != 1.0
    g_lu = ulp_gap(lu, olu)
    logz_dev, logz_ora = t1 / (NDIM - 1), ot1 / (NDIM - 1)
    g_t1 = np.abs(t1 - ot1) / np.spacing(np.abs(np.where(nz, ot1, 1.0)))
    frac_lu, frac_t1 = float((lu != olu).mean()), float((t1 != ot1).mean())
    max_lu, max_t1 = float(g_lu.max()), float(g_t1[nz].max())
    assert max_lu <= 2.0 and max_t1 <= 3.0, (max_lu, max_t1)
    # absolute gaps (what enters the comparison lhs >= lu)
    abs_lu, abs_t1 = np.abs(lu - olu), np.abs(t1 - ot1)
    gap = abs_lu + abs_t1                                   # per draw: the two sides can disagree only if |lhs - lu| <= gap
    # A flip needs the oracle's (lhs - lu) to land inside [-gap, gap].  lhs - lu = t1 + (p1 - p0) - lu where lu = log u is
    # continuous with density e^lu <= 1 on (-inf, 0): P(flip | draw) <= 2 * gap * max density = 2 * gap.  (With p1 - p0 random
    # as well the bound only gets smaller.)  Expected flips per 10^9 walker-steps:
    flips_per_1e9 = float(2.0 * gap.mean() * 1e9)
    # ... and directly: the number of these 10^7 draws whose decision WOULD differ for the C2-typical log-pdf differences
    rng = np.random.default_rng(7)
    dlp = -np.abs(rng.standard_normal(N)) * 4.0             # p1 - p0 at 32 dims, a = 2: mostly negative, scale of a few units
    dev = (t1 + dlp) >= lu
    ora = (ot1 + dlp) >= olu
    observed = int((dev != ora).sum())
    with capsys.disabled():
        print(f"\n[accept margin] {N} draws: lu differs in {frac_lu:.3%} (max {max_lu:.2f} ulp), t1 in {frac_t1:.3%} (max {max_t1:.2f} ulp); "
              f"mean absolute gap {gap.mean():.3e}, max {gap.max():.3e}; bound on flipped accept decisions: {flips_per_1e9:.2e} per 10^9 walker-steps; "
              f"observed on synthetic log-pdf differences: {observed} of {N}")
    assert flips_per_1e9 < 1e-3          # < one flipped decision per 10^12 walker-steps (a C2 job is 6.6e8)
    assert observed == 0
    assert abs(float(dev.mean()) - float(ora.mean())) == 0.0


def test_accept_terms_argument_checks(kmc):
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    assert L.kmc_debug_accept_terms(1, 0, 0, 4, 8, 1.0, 2, 0, None, None, None, None) == _lib.ERR_BAD_ARG


def test_logpdf_gap_of_the_lane_striped_sum(kmc, oracle, capsys):
    """The other inexact term of the accept test: p1 (and p0, an earlier p1).  The vector kernels sum a row's terms lane-striped
    + butterfly, the oracle in index order: same terms, different rounding.  Measured on C2-shaped rows after 60 generations
    (positions are bit-identical, so the two log-pdf arrays are sums of the SAME numbers in two orders)."""
    nw, nd, G, seed = 8192, 32, 60, 5
    th = np.random.default_rng(3).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 0, 1, 2.0, seed) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        pos, lp = s.positions(), s.logp()
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, 0, 1, 2.0, seed, nthreads=8), th, store_chain=False)
    np.testing.assert_array_equal(pos, ref["final_pos"])
    gap = np.abs(lp - ref["final_logp"])
    rel = gap / np.spacing(np.abs(ref["final_logp"]))
    flips_per_1e9 = float(2.0 * (2.0 * gap.mean()) * 1e9)         # p1 and p0 both carry such a gap; density of log u <= 1
    with capsys.disabled():
        print(f"\n[accept margin] log-pdf, {nw} x {nd} Gaussian rows: differs in {float((gap > 0).mean()):.1%} of the walkers, max {rel.max():.1f} ulp "
              f"({gap.max():.2e} absolute), mean {gap.mean():.2e}; bound on flipped accept decisions from this term: {flips_per_1e9:.2e} per 10^9 walker-steps")
    assert rel.max() <= 16 and flips_per_1e9 < 1e-2
