"""GPU: device-side make_theta0s (kmc_sampler_init_ball) and checkpoint / resume (kmc_sampler_set_state)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
