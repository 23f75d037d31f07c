"""CPU: the oracle's seeded initial ball (kmco_init_ball -- the restatement of reference src/samplers.jl:311-349 that the
device-side kmc_sampler_init_ball is compared with) against its committed fixtures, against an independent numpy
restatement of its random-stream contract, and against the reference's rules (first admissible try is kept, the ball
shrinks by 1, 1/2, 1/8, 1/64, ... within a walker, :324-341)."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = sorted(glob.glob(os.path.join(HERE, "golden", "init_ball", "*.npz")))


def load(path):
    z = dict(np.load(path))
    for k in ("density", "nwalkers", "ndim", "halving_steps", "ntries", "seed", "nfail"):
        z[k] = int(z[k])
    return z


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_oracle_reproduces_init_ball_fixture(oracle, path):
    z = load(path)
    r = oracle.init_ball(z["density"], list(z["params"]), z["theta0"], z["radius"], z["nwalkers"], z["ndim"], seed=z["seed"],
                         halving_steps=z["halving_steps"], ntries=z["ntries"])
    assert r["nfail"] == z["nfail"]
    np.testing.assert_array_equal(r["attempts"], z["attempts"])
    np.testing.assert_allclose(r["pos"], z["pos"], rtol=1e-14, atol=0)
    ok = z["attempts"] > 0
    np.testing.assert_allclose(r["logp"][ok], z["logp"][ok], rtol=1e-13)
    assert np.all(np.isneginf(r["logp"][~ok]))


def _normals(oracle, seed, attempt, walker, ndim):
    """The contract of kmc_oracle.c: kmco_init_ball, restated with numpy on top of the Philox block function."""
    out = np.zeros(ndim)
    for pair in range((ndim + 1) // 2):
        w = oracle.philox4x32_10((attempt, pair, walker & 0xFFFFFFFF, walker >> 32), ((seed & 0xFFFFFFFF) ^ 0x42414C4C, seed >> 32))
        u1 = (((w[0] << 20) | (w[1] >> 12)) + 0.5) * 2.0 ** -52
        u2 = (w[2] + 0.5) * 2.0 ** -32
        r = np.sqrt(-2.0 * np.log(u1))
        out[2 * pair] = r * np.cos(2.0 * np.pi * u2)
        if 2 * pair + 1 < ndim:
            out[2 * pair + 1] = r * np.sin(2.0 * np.pi * u2)
    return out


def test_init_ball_follows_the_reference_rules(oracle):
    """Walker by walker, in plain Python: tries in order, the FIRST admissible one is kept (src/samplers.jl:336-341),
    ntries per ball size (:327), the radius scaled by the compounding 1/2^(k-1) of :326 -- restarted per walker."""
    z = load(os.path.join(HERE, "golden", "init_ball", "expo_shrink_200x8.npz"))
    nd, nt = z["ndim"], z["ntries"]
    for w in range(0, z["nwalkers"], 7):
        shrink, attempt, found = 1.0, 0, None
        for k in range(1, z["halving_steps"] + 1):
            shrink *= 1.0 / 2 ** (k - 1)
            for _ in range(nt):
                x = z["theta0"] + _normals(oracle, z["seed"], attempt, w, nd) * (z["radius"] * shrink)
                attempt += 1
                if oracle.logpdf(z["density"], z["params"], x) > -np.inf:
                    found = x
                    break
            if found is not None:
                break
        assert found is not None and attempt == z["attempts"][w]
        np.testing.assert_allclose(found, z["pos"][w], rtol=1e-13, atol=1e-15)
    assert (z["attempts"] > 2 * nt).any() and (z["attempts"] <= nt).any()      # the fixture does exercise the shrinking


def test_init_ball_is_a_pure_function_of_the_global_walker_index(oracle):
    """Rows [a, b) of a whole-ensemble ball == the ball of a shard that starts at walker a (how P2P shards call it)."""
    full = oracle.init_ball(oracle.EXPONENTIAL, [1.0], 0.02, 0.1, 96, 3, seed=11)
    part = oracle.init_ball(oracle.EXPONENTIAL, [1.0], 0.02, 0.1, 40, 3, seed=11, walker0=32)
    np.testing.assert_array_equal(part["pos"], full["pos"][32:72])
    np.testing.assert_array_equal(part["attempts"], full["attempts"][32:72])


def test_init_ball_statistics(oracle):
    """:328-332 theta0 .+ randn(npara) .* ball_radius: unit normals, independent across walkers and dimensions."""
    from scipy import stats
    nd = 6
    th, rad = np.linspace(-1, 1, nd), np.linspace(0.05, 0.3, nd)
    r = oracle.init_ball(oracle.GAUSSIAN_ISO, [0.0, 1.0], th, rad, 40000, nd, seed=2)
    z = (r["pos"] - th) / rad
    assert np.all(r["attempts"] == 1)
    assert np.abs(z.mean(axis=0)).max() < 0.03 and np.abs(z.std(axis=0) - 1).max() < 0.02
    assert np.abs(np.corrcoef(z.T) - np.eye(nd)).max() < 0.03
    for d in range(nd):
        assert stats.kstest(z[:, d], "norm").pvalue > 1e-4
    # conditioned on pdf > -Inf (exponential: x >= 0): accepted points are never outside the support
    e = oracle.init_ball(oracle.EXPONENTIAL, [1.0], 0.02, 0.1, 5000, 2, seed=3)
    assert e["nfail"] == 0 and e["pos"].min() >= 0.0
    assert 0.6 < (e["attempts"] > 1).mean() < 0.75          # P(both coordinates >= 0 at the first try) = Phi(0.2)^2 = 0.335
