"""Writes tests/golden/host/handmade.json: small HAND-DERIVED cases for the host-side functions of the emcee path --
`squash_walkers` (reference src/samplers.jl:372-428) and `make_theta0s` (src/samplers.jl:311-349).

Every expected value below was worked out by reading the reference's code, line by line, on paper; nothing here
imports the package under test or the oracle.  The only arithmetic this script does is `theta0 + normal * radius`
for make_theta0s, with the radius of each try taken from the hand-written schedules (the normals are the first
draws of numpy's PCG64 generator for the stated seed and are stored in the fixture, so a change of numpy's stream
is detected by the test instead of silently moving the expectation).
Re-run: python tests/golden/host/make_handmade.py"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NINF = "-inf"

# ---------------------------------------------------------------------------------------------------------
# squash_walkers.  thetas[w][k]: sample k of walker w.
# ---------------------------------------------------------------------------------------------------------
T3 = [[11.0, 12.0], [21.0, 22.0], [31.0, 32.0]]
L3 = [[-1.1, -1.2], [-2.1, -2.2], [-3.1, -3.2]]
B3 = [["a1", "a2"], ["b1", "b2"], ["c1", "c2"]]
squash = [
    dict(name="default_order_walker_major",
         derivation=":395 walkers2keep = 1:3.  :398 t = copy(thetas[1]) = [11,12]; :399 append thetas[2], thetas[3] -> walker-major. "
                    ":403-405 the same for logdensities.  :408 blobs == nothing -> b = nothing.  :427 mean(accept_ratio) = (0.5+0.3+0.4)/3 = 0.4.",
         thetas=T3, accept_ratio=[0.5, 0.3, 0.4], logdensities=L3, blobs=None, kwargs={},
         expect=dict(thetas=[11.0, 12.0, 21.0, 22.0, 31.0, 32.0], accept=0.4, logdensities=[-1.1, -1.2, -2.1, -2.2, -3.1, -3.2], blobs=None)),
    dict(name="order_true_sample_major",
         derivation=":416-418 nc = 3, ns = 2, keys = vcat([1,2],[1,2],[1,2]) = [1,2,1,2,1,2]; sortperm is stable, so perm = [1,3,5,2,4,6] "
                    "(1-based): the first samples of walkers 1,2,3, then their second samples.  :419-425 b, l, t are all indexed by perm.",
         thetas=T3, accept_ratio=[0.5, 0.3, 0.4], logdensities=L3, blobs=B3, kwargs=dict(order=True),
         expect=dict(thetas=[11.0, 21.0, 31.0, 12.0, 22.0, 32.0], accept=0.4, logdensities=[-1.1, -2.1, -3.1, -1.2, -2.2, -3.2],
                     blobs=["a1", "b1", "c1", "a2", "b2", "c2"])),
    dict(name="blobs_merged_by_append",
         derivation=":411 b = deepcopy(blobs[1]); :412 merge_blobs! = append! of blobs[2], blobs[3] -> [a1,a2,b1,b2,c1,c2]; the caller's blobs[1] "
                    "keeps its two entries (deepcopy).  No logdensities passed -> l = nothing (:401-402).",
         thetas=T3, accept_ratio=[0.5, 0.3, 0.4], logdensities=None, blobs=B3, kwargs={},
         expect=dict(thetas=[11.0, 12.0, 21.0, 22.0, 31.0, 32.0], accept=0.4, logdensities=None, blobs=["a1", "a2", "b1", "b2", "c1", "c2"])),
    dict(name="drop_rule_is_less_or_equal",
         derivation=":384 ma = median([1,1,0,1]) = 1; sa = std (n-1 normalisation): mean 0.75, squared deviations 3 x 0.0625 + 0.5625 = 0.75, / 3 = 0.25, "
                    "sqrt = 0.5 (all exact in binary).  drop_fact = 2 (default): threshold ma - 2 sa = 0.  :387 accept_ratio[3] = 0 <= 0 -> walker 3 is dropped "
                    "(a strict < would keep it).  Kept 1,2,4: t = their samples walker-major; :427 mean of the kept ratios = 1.",
         thetas=[[1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [7.0, 8.0]], accept_ratio=[1.0, 1.0, 0.0, 1.0],
         logdensities=[[-1.0, -2.0], [-3.0, -4.0], [-5.0, -6.0], [-7.0, -8.0]], blobs=None, kwargs=dict(drop_low_accept_ratio=True),
         expect=dict(thetas=[1.0, 2.0, 3.0, 4.0, 7.0, 8.0], accept=1.0, logdensities=[-1.0, -2.0, -3.0, -4.0, -7.0, -8.0], blobs=None)),
    dict(name="drop_fact_decides",
         derivation="accept_ratio = [0.30, 0.32, 0.02, 0.34]: median = (0.30 + 0.32)/2 = 0.31; mean 0.245, deviations 0.055, 0.075, -0.225, 0.095, squares "
                    "0.003025 + 0.005625 + 0.050625 + 0.009025 = 0.0683, / 3 = 0.022767, std = 0.150887.  drop_fact = 2: threshold 0.31 - 0.301773 = 0.008227 < 0.02 "
                    "-> nobody is dropped, mean = 0.245.",
         thetas=[[1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [7.0, 8.0]], accept_ratio=[0.30, 0.32, 0.02, 0.34], logdensities=None, blobs=None,
         kwargs=dict(drop_low_accept_ratio=True, drop_fact=2),
         expect=dict(thetas=[1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0], accept=0.245, logdensities=None, blobs=None)),
    dict(name="drop_and_order_with_blobs",
         derivation="same ratios, drop_fact = 1.5: threshold 0.31 - 0.226330 = 0.083670 >= 0.02 -> walker 3 dropped, kept 1,2,4 (nc = 3).  Walker-major "
                    "t = [1,2,3,4,7,8]; order=true: ns = length(thetas[1]) = 2, perm = [1,3,5,2,4,6] -> [1,3,7,2,4,8]; blobs alike.  mean = (0.30+0.32+0.34)/3 = 0.32.",
         thetas=[[1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [7.0, 8.0]], accept_ratio=[0.30, 0.32, 0.02, 0.34],
         logdensities=[[-1.0, -2.0], [-3.0, -4.0], [-5.0, -6.0], [-7.0, -8.0]], blobs=[["p", "q"], ["r", "s"], ["t", "u"], ["v", "w"]],
         kwargs=dict(drop_low_accept_ratio=True, drop_fact=1.5, order=True),
         expect=dict(thetas=[1.0, 3.0, 7.0, 2.0, 4.0, 8.0], accept=0.32, logdensities=[-1.0, -3.0, -7.0, -2.0, -4.0, -8.0], blobs=["p", "r", "v", "q", "s", "w"])),
    dict(name="vector_walkers",
         derivation="thetas[w][k] is a 2-vector (ndim = 2): :398-399 concatenate the walkers' vectors of samples; each sample stays a 2-vector. order=true "
                    "interleaves them by sample index as above.",
         thetas=[[[1.0, -1.0], [2.0, -2.0]], [[3.0, -3.0], [4.0, -4.0]]], accept_ratio=[0.25, 0.75], logdensities=None, blobs=None, kwargs=dict(order=True),
         expect=dict(thetas=[[1.0, -1.0], [3.0, -3.0], [2.0, -2.0], [4.0, -4.0]], accept=0.5, logdensities=None, blobs=None)),
]

# ---------------------------------------------------------------------------------------------------------
# make_theta0s.  The pdf of a case rejects (returns -Inf for) exactly the candidate values listed in `reject`.
# schedule rows: [try number (1-based, = index of the normal draw), walker i, halving step k, radius, accepted?]
# ---------------------------------------------------------------------------------------------------------
SEED = 2024
NORMALS = np.random.default_rng(SEED).standard_normal(12)


def scalar_case(name, derivation, theta0, radius0, nwalkers, halving, ntries, schedule, api_raises=False):
    cand = {t: theta0 + NORMALS[t - 1] * r for t, _, _, r, _ in schedule}          # :329 theta0 .+ randn().*ball_radius
    return dict(name=name, derivation=derivation, theta0=theta0, ball_radius=radius0, nwalkers=nwalkers,
                ball_radius_halfing_steps=halving, ntries=ntries, seed=SEED, schedule=schedule,
                reject=[cand[t] for t, _, _, _, ok in schedule if not ok],
                expect=[cand[t] for t, _, _, _, ok in schedule if ok], api_raises=api_raises)


make = [
    scalar_case(
        "compounding_shrink_is_never_reset",
        ":326 ball_radius *= 1/2^(k-1) multiplies the CURRENT radius and nothing ever resets it.  Start r = 1, ntries = 2, 3 halving steps.  "
        "Walker 1: k=1 r = 1*1 = 1: tries 1, 2 rejected; :343 length 0 != 1; k=2 r = 1/2: try 3 rejected, try 4 accepted (pushed), break; :343 length 1 == 1, break.  "
        "Walker 2: k=1 r = 0.5*1 = 0.5 (carried over): try 5 accepted.  Walker 3: k=1 r = 0.5: tries 6, 7 rejected; k=2 r = 0.25: tries 8, 9 rejected; "
        "k=3 r = 0.25/4 = 0.0625: try 10 accepted.  One randn() per try (npara == 1, :328-329).",
        0.0, 1.0, 3, 3, 2,
        [[1, 1, 1, 1.0, False], [2, 1, 1, 1.0, False], [3, 1, 2, 0.5, False], [4, 1, 2, 0.5, True], [5, 2, 1, 0.5, True],
         [6, 3, 1, 0.5, False], [7, 3, 1, 0.5, False], [8, 3, 2, 0.25, False], [9, 3, 2, 0.25, False], [10, 3, 3, 0.0625, True]]),
    scalar_case(
        "first_try_each",
        "Every first try is admissible: r stays 1 (k = 1 multiplies by 1/2^0 = 1); walker i takes draw i.  theta0 = 0.5 shifts every candidate.",
        0.5, 1.0, 4, 7, 100,
        [[1, 1, 1, 1.0, True], [2, 2, 1, 1.0, True], [3, 3, 1, 1.0, True], [4, 4, 1, 1.0, True]]),
    scalar_case(
        "failed_walker_quirk",
        "ntries = 2, 3 halving steps, walker 1 never finds a point: k=1 r=1 (tries 1,2), k=2 r=0.5 (3,4), k=3 r=0.125 (5,6) all rejected.  :344 reads the OUTER "
        "j = 0 (the loop variable of :327 is local), so `j==ntries` is false and error() is never reached: nothing is pushed for walker 1.  Walker 2: k=1 "
        "r = 0.125: try 7 accepted, pushed -> length 1 != i = 2, so :343 does not break; k=2 r = 0.0625: try 8 accepted, pushed -> length 2 == 2, break.  "
        "Walker 3: k=1 r = 0.0625: try 9 accepted, length 3 == 3.  The reference returns these three values (two of them from walker 2's loop); the "
        "product's host make_theta0s deliberately raises instead when a walker finds nothing (DESIGN.md section 1).",
        0.0, 1.0, 3, 3, 2,
        [[1, 1, 1, 1.0, False], [2, 1, 1, 1.0, False], [3, 1, 2, 0.5, False], [4, 1, 2, 0.5, False], [5, 1, 3, 0.125, False], [6, 1, 3, 0.125, False],
         [7, 2, 1, 0.125, True], [8, 2, 2, 0.0625, True], [9, 3, 1, 0.0625, True]], api_raises=True),
]

# vector walkers: theta0 = [1, -1], scalar ball_radius 0.5 -> ones(2) * 0.5 (:316-318); randn(2) per try (:331): try t uses draws 2t-1, 2t
vec_theta0 = [1.0, -1.0]
vec_cands = {t: [vec_theta0[0] + NORMALS[2 * t - 2] * r, vec_theta0[1] + NORMALS[2 * t - 1] * r] for t, r in [(1, 0.5), (2, 0.5), (3, 0.5)]}
make.append(dict(
    name="vector_walkers_scalar_radius_broadcast",
    derivation=":315 npara = 2; :316-318 ball_radius = ones(2)*0.5; ntries = 2.  Walker 1: k=1 r = 0.5: try 1 rejected, try 2 accepted.  Walker 2: k=1 r = 0.5: "
               "try 3 accepted.  Try t consumes randn(2) = draws 2t-1, 2t.",
    theta0=vec_theta0, ball_radius=0.5, nwalkers=2, ball_radius_halfing_steps=7, ntries=2, seed=SEED,
    schedule=[[1, 1, 1, 0.5, False], [2, 1, 1, 0.5, True], [3, 2, 1, 0.5, True]],
    reject=[vec_cands[1]], expect=[vec_cands[2], vec_cands[3]], api_raises=False))

json.dump(dict(about="hand-derived expectations for squash_walkers (src/samplers.jl:372-428) and make_theta0s (src/samplers.jl:311-349); see make_handmade.py",
               numpy_seed=SEED, normals=[float(v) for v in NORMALS], squash_walkers=squash, make_theta0s=make),
          open(os.path.join(HERE, "handmade.json"), "w"), indent=1)
print("wrote", os.path.join(HERE, "handmade.json"))
