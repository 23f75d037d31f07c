"""Host logic (no GPU): `squash_walkers` and `make_theta0s` of the product (kissmcmc_jl_amd.api) AND of the oracle
(oracle/host.py) against small HAND-DERIVED expectations (tests/golden/host/handmade.json, worked out on paper from
reference src/samplers.jl:372-428 and :311-349 -- each case carries its derivation).  The two Python restatements were
written by the same hand; these fixtures are what keeps their agreement from being a self-comparison."""
import copy
import json
import os

import numpy as np
import pytest

from oracle import host as ohost

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "host", "handmade.json")))
SQ = FIX["squash_walkers"]
MK = FIX["make_theta0s"]


def test_numpy_stream_has_not_moved():
    """The make_theta0s expectations are literal numbers built on the first normals of default_rng(seed)."""
    np.testing.assert_array_equal(np.random.default_rng(FIX["numpy_seed"]).standard_normal(len(FIX["normals"])), np.array(FIX["normals"]))


@pytest.mark.parametrize("case", SQ, ids=[c["name"] for c in SQ])
def test_squash_walkers_product_matches_hand_derivation(kmc, case):
    blobs_in = copy.deepcopy(case["blobs"])
    t, a, l, b = kmc.squash_walkers(np.array(case["thetas"]), case["accept_ratio"],
                                    None if case["logdensities"] is None else np.array(case["logdensities"]),
                                    blobs_in, verbose=False, **case["kwargs"])
    e = case["expect"]
    np.testing.assert_array_equal(t, np.array(e["thetas"]))
    assert a == pytest.approx(e["accept"], rel=1e-15)
    if e["logdensities"] is None:
        assert l is None
    else:
        np.testing.assert_array_equal(l, np.array(e["logdensities"]))
    assert b == e["blobs"]
    assert blobs_in == case["blobs"]                         # :411 deepcopy: the caller's blobs are not modified


@pytest.mark.parametrize("case", SQ, ids=[c["name"] for c in SQ])
def test_squash_walkers_oracle_matches_hand_derivation(case):
    kw = dict(case["kwargs"])
    t, a, l, b = ohost.squash_walkers(case["thetas"], case["accept_ratio"], case["logdensities"], None, **kw)   # the oracle restates no blobs
    e = case["expect"]
    assert [list(np.atleast_1d(x)) for x in t] == [list(np.atleast_1d(x)) for x in e["thetas"]]
    assert a == pytest.approx(e["accept"], rel=1e-15)
    assert l == e["logdensities"]


def _pdf_rejecting(values):
    vals = [np.atleast_1d(np.array(v, dtype=np.float64)) for v in values]

    def pdf(x):
        x = np.atleast_1d(np.asarray(x, dtype=np.float64))
        return -np.inf if any(np.array_equal(x, v) for v in vals) else -float(np.sum(x * x))
    return pdf


@pytest.mark.parametrize("case", MK, ids=[c["name"] for c in MK])
def test_make_theta0s_schedule_is_consistent(case):
    """The literal expectations are exactly theta0 + normal(try) * radius(try) of the hand-written schedule."""
    n = np.array(FIX["normals"])
    th = np.array(case["theta0"], dtype=np.float64)
    acc = [(t, r) for t, _, _, r, ok in case["schedule"] if ok]
    for (t, r), want in zip(acc, case["expect"]):
        got = th + (n[t - 1] if th.ndim == 0 else n[2 * t - 2:2 * t]) * r
        np.testing.assert_array_equal(np.atleast_1d(got), np.atleast_1d(np.array(want)))
    # the radii follow :326 -- within a walker the radius of step k is the previous radius times 1/2^(k-1), never reset
    r_now, prev_walker, prev_k = case["ball_radius"] if np.ndim(case["ball_radius"]) == 0 else None, 0, 0
    for _, i, k, r, _ in case["schedule"]:
        if (i, k) != (prev_walker, prev_k):
            ks = range(1, k + 1) if i != prev_walker else range(prev_k + 1, k + 1)
            for kk in ks:
                r_now = r_now * (1 / 2 ** (kk - 1))
            prev_walker, prev_k = i, k
        assert r == r_now


@pytest.mark.parametrize("case", MK, ids=[c["name"] for c in MK])
def test_make_theta0s_oracle_matches_hand_derivation(case):
    got = ohost.make_theta0s(case["theta0"] if np.ndim(case["theta0"]) == 0 else np.array(case["theta0"]), case["ball_radius"],
                             _pdf_rejecting(case["reject"]), case["nwalkers"], np.random.default_rng(case["seed"]),
                             ball_radius_halfing_steps=case["ball_radius_halfing_steps"], ntries=case["ntries"])
    assert len(got) == len(case["expect"])
    for g, w in zip(got, case["expect"]):
        np.testing.assert_array_equal(np.atleast_1d(g), np.atleast_1d(np.array(w)))


@pytest.mark.parametrize("case", MK, ids=[c["name"] for c in MK])
def test_make_theta0s_product_matches_hand_derivation(kmc, case):
    args = (case["theta0"] if np.ndim(case["theta0"]) == 0 else np.array(case["theta0"]), case["ball_radius"],
            _pdf_rejecting(case["reject"]), case["nwalkers"])
    kw = dict(ball_radius_halfing_steps=case["ball_radius_halfing_steps"], ntries=case["ntries"], rng=np.random.default_rng(case["seed"]))
    if case["api_raises"]:
        # deliberate deviation: the reference silently returns entries pushed by LATER walkers'
        # loops when one walker finds nothing (its error() at :345 is unreachable); the product raises
        with pytest.raises(RuntimeError, match="Could not find suitable initial theta"):
            kmc.make_theta0s(*args, **kw)
        return
    got = kmc.make_theta0s(*args, **kw)
    np.testing.assert_array_equal(got, np.array(case["expect"]))
    assert got.shape == ((case["nwalkers"],) if np.ndim(case["theta0"]) == 0 else (case["nwalkers"], len(case["theta0"])))
