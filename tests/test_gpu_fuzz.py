"""GPU: a seeded random sweep of sampler configurations against the oracle -- shapes nobody picked by hand (odd ndim,
ensembles just above ndim + 2, active halves that do not fill a wave, thinning that does not divide the run), every
density of the menu, every launch mode, float / double rows excepted (f32 has its own statistical tests), with the job
cut into random run() pieces.  Same bar as test_gpu_parity.py: identical counters, bit-identical positions and chains,
log-pdfs to 1e-12, moments to 1e-11."""
import numpy as np
import pytest

import kmcenv

pytestmark = pytest.mark.gpu

import os

N_TRIALS = int(os.environ.get("KMC_FUZZ_TRIALS", 120))          # (a longer hunt: KMC_FUZZ_TRIALS=4000 KMC_FUZZ_BASE=...)
BASE = int(os.environ.get("KMC_FUZZ_BASE", 9000))


def _draw_config(rng):
    name = str(rng.choice(["gauss", "gauss_shift", "expo", "rosen", "lognormal", "mvn2"]))
    if name == "mvn2":
        nd = 2
    elif name == "rosen":
        nd = int(rng.choice([2, 3, 5, 8, 17, 32, 33, 64, 100, 130]))
    else:
        nd = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 15, 16, 31, 32, 33, 63, 64, 65, 127, 128, 200, 257, 600]))
    lo = nd + 2 + (nd % 2)                                             # even and >= ndim + 2
    nw = int(rng.choice([lo, lo + 2, 2 * (nd + 3), 64, 100, 130, 256, 258, 1000, 1026, 2050, 4096]))
    nw = max(nw, lo)
    nw += nw % 2
    if nd >= 200:
        nw = min(nw, 1026) if nw >= lo else lo
        nw = max(nw, lo)
    G = int(rng.integers(3, 140 if nd < 200 else 12))
    nburn = int(rng.integers(0, G))
    nthin = int(rng.choice([1, 1, 2, 3, 5]))
    a = float(rng.choice([2.0, 2.0, 1.3, 3.5]))
    launch = str(rng.choice(["", "graph", "updated", "eager"]))
    resident = bool(rng.integers(0, 2))
    return name, nw, nd, G, nburn, nthin, a, launch, resident


@pytest.mark.parametrize("trial", range(N_TRIALS))
def test_random_configuration_equals_the_oracle(kmc, oracle, monkeypatch, trial):
    rng = np.random.default_rng(BASE + trial)
    name, nw, nd, G, nburn, nthin, a, launch, resident = _draw_config(rng)
    dens = {
        "gauss": (kmc.GaussianIso(0.0, 1.0), oracle.GAUSSIAN_ISO, [0.0, 1.0]),
        "gauss_shift": (kmc.GaussianIso(-5.0, 3.0), oracle.GAUSSIAN_ISO, [-5.0, 3.0]),
        "expo": (kmc.Exponential(1.0), oracle.EXPONENTIAL, [1.0]),
        "rosen": (kmc.Rosenbrock(1.0, 100.0, 20.0), oracle.ROSENBROCK, [1.0, 100.0, 20.0]),
        "lognormal": (kmc.LogNormal(0.0, 1.0), oracle.LOGNORMAL, [0.0, 1.0]),
        "mvn2": (kmc.MvNormal2([0.5, -0.25], [[0.47, 0.2], [0.2, 7.0]]), oracle.MVNORMAL2,
                 kmc.MvNormal2([0.5, -0.25], [[0.47, 0.2], [0.2, 7.0]]).params()),
    }
    pdf, did, params = dens[name]
    if name in ("expo", "lognormal"):
        th = 0.55 + 0.1 * np.abs(rng.standard_normal((nw, nd)))
    elif name == "gauss_shift":
        th = -4.0 + 0.1 * rng.standard_normal((nw, nd))
    else:
        th = 0.1 * rng.standard_normal((nw, nd))
    if launch:
        monkeypatch.setenv("KMC_LAUNCH", launch)
    if not resident:
        kmcenv.no_resident(monkeypatch)
    seed = int(rng.integers(1, 2 ** 40))
    label = f"trial {trial}: {name} {nw}x{nd} G={G} nburn={nburn} nthin={nthin} a={a} launch={launch or 'auto'} resident={resident}"
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, nthin, a, seed), th)
    assert ref["status"] == 0, label
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, a, seed, store_chain=True, store_logp=True, moments=True) as s:
        s.set_positions(th)
        left = G
        while left > 0:                                  # the job in random pieces
            n = int(min(left, rng.choice([1, 2, 7, 63, 64, 65, 200])))
            s.run(n)
            left -= n
            if rng.random() < 0.3:
                s.sync()
        s.sync()
        chain, clogp = s.chain()
        np.testing.assert_array_equal(s.naccept(), ref["naccept"], err_msg=label)
        np.testing.assert_array_equal(s.positions(), ref["final_pos"], err_msg=label)
        np.testing.assert_array_equal(chain, ref["chain"], err_msg=label)
        tol = 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"]))
        assert np.all(np.abs(s.logp() - ref["final_logp"]) <= tol), label
        tol = 1e-12 * np.maximum(1.0, np.abs(ref["chain_logp"]))
        assert np.all(np.abs(clogp - ref["chain_logp"]) <= tol), label
        msum, msq, n = s.moments()
        assert n == ref["nmoment"], label
        np.testing.assert_allclose(msum, ref["sum"], rtol=1e-11, atol=1e-9, err_msg=label)
        np.testing.assert_allclose(msq, ref["sumsq"], rtol=1e-11, atol=1e-9, err_msg=label)
        cw, lw = s.chain(by_walker=True)
        np.testing.assert_array_equal(cw, chain.transpose(1, 0, 2), err_msg=label)
        np.testing.assert_array_equal(lw, clogp.T, err_msg=label)


@pytest.mark.parametrize("trial", range(max(8, N_TRIALS // 6)))
def test_random_metropolis_configuration_equals_the_oracle(kmc, oracle, trial):
    """Many-chain Metropolis (src/samplers.jl:59-128): random chain counts, dimensions (register geometries and the
    from-memory kernel), steps, thinning; counters identical, states within 1e-11 (the tolerance of test_gpu_metropolis.py:
    device log / sin / cos differ from glibc's by <= 1 ulp in the Box-Muller normals), both chain layouts."""
    from kissmcmc_jl_amd.metropolis import run_chains
    rng = np.random.default_rng(BASE + 77000 + trial)
    name = str(rng.choice(["gauss", "expo", "rosen", "lognormal"]))
    nd = int(rng.choice([2, 3, 5, 8, 16, 33]) if name == "rosen" else rng.choice([1, 2, 3, 4, 7, 8, 9, 16, 17, 31, 32, 33, 48]))
    nc = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 777, 4096, 5001]))
    niter = int(rng.integers(1, 160))
    nburn = int(rng.integers(0, niter))
    nthin = int(rng.choice([1, 1, 2, 5]))
    seed = int(rng.integers(1, 2 ** 40))
    pdf, did, params = {"gauss": (kmc.GaussianIso(0.1, 1.3), oracle.GAUSSIAN_ISO, [0.1, 1.3]), "expo": (kmc.Exponential(1.0), oracle.EXPONENTIAL, [1.0]),
                        "rosen": (kmc.Rosenbrock(), oracle.ROSENBROCK, [1.0, 100.0, 20.0]), "lognormal": (kmc.LogNormal(0.0, 1.0), oracle.LOGNORMAL, [0.0, 1.0])}[name]
    th = 0.6 + 0.1 * np.abs(rng.standard_normal((nc, nd))) if name in ("expo", "lognormal") else 0.3 * rng.standard_normal((nc, nd))
    step = rng.uniform(0.05, 0.9, nd) if rng.random() < 0.5 else float(rng.uniform(0.05, 0.9))
    label = f"trial {trial}: {name} {nc} chains x {nd}, niter={niter} nburn={nburn} nthin={nthin}"
    r = run_chains(pdf, kmc.GaussianStep(step), th, niter, nburn, nthin, seed, moments=True, by_chain=bool(trial % 2))
    ref = oracle.metropolis(did, params, th, step, niter, nburn, nthin, seed)
    np.testing.assert_array_equal(r["naccept"], ref["naccept"], err_msg=label)
    chain = r["chain"].transpose(1, 0, 2) if trial % 2 else r["chain"]
    clogp = r["chain_logp"].T if trial % 2 else r["chain_logp"]
    np.testing.assert_allclose(chain, ref["chain"], rtol=1e-11, atol=1e-11, err_msg=label)
    np.testing.assert_allclose(clogp, ref["chain_logp"], rtol=1e-10, atol=1e-10, err_msg=label)
    np.testing.assert_allclose(r["final_pos"], ref["final_pos"], rtol=1e-11, atol=1e-11, err_msg=label)
    np.testing.assert_allclose(r["chain_sum"], ref["chain_sum"], rtol=1e-10, atol=1e-10, err_msg=label)


@pytest.mark.parametrize("trial", range(max(6, N_TRIALS // 10)))
def test_random_logical_shards_equal_the_oracle(kmc, oracle, trial):
    """Walker sharding (replica form: P samplers with shard_rank r on one position buffer, the device-local stand-in for the
    all-gather): random shard counts, shapes and densities; the sharded run is the oracle's unsharded run bit for bit."""
    import torch
    rng = np.random.default_rng(BASE + 33000 + trial)
    P = int(rng.choice([2, 4, 8]))
    nd = int(rng.choice([2, 4, 8, 16, 32, 64, 100]))                   # (bind_positions: even ndim)
    per = int(rng.choice([2, 3, 8, 33, 64]))                           # active walkers per shard and half
    nw = max(2 * P * per, ((nd + 2 + 2 * P - 1) // (2 * P)) * 2 * P)
    name = str(rng.choice(["gauss", "rosen", "expo"]))
    pdf, did, params = {"gauss": (kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0]), "rosen": (kmc.Rosenbrock(), oracle.ROSENBROCK, [1.0, 100.0, 20.0]),
                        "expo": (kmc.Exponential(), oracle.EXPONENTIAL, [1.0])}[name]
    G = int(rng.integers(2, 40)); nburn = int(rng.integers(0, G)); seed = int(rng.integers(1, 2 ** 40))
    th = 0.6 + 0.1 * np.abs(rng.standard_normal((nw, nd))) if name == "expo" else 0.3 * rng.standard_normal((nw, nd))
    pos = torch.empty((nw, nd), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    shards = []
    try:
        for r in range(P):
            s = kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed, moments=True, use_graph=False, shard_rank=r, shard_count=P)
            s.bind_positions(pos.data_ptr())
            s.set_stream(stream)
            s.set_positions(th)
            shards.append(s)
        for g in range(G):
            for half in (0, 1):
                for s in shards:
                    s.half_step(half)
        torch.cuda.synchronize()
        nacc = sum(s.naccept() for s in shards)
        n = sum(s.moments()[2] for s in shards)
        S = sum(s.moments()[0] for s in shards)
    finally:
        for s in shards:
            s.close()
    label = f"trial {trial}: {name} {nw}x{nd} in {P} shards, G={G} nburn={nburn}"
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, 1, 2.0, seed), th, store_chain=False)
    np.testing.assert_array_equal(pos.cpu().numpy(), ref["final_pos"], err_msg=label)
    np.testing.assert_array_equal(nacc, ref["naccept"], err_msg=label)
    assert n == ref["nmoment"], label
    np.testing.assert_allclose(S, ref["sum"], rtol=1e-11, atol=1e-9, err_msg=label)


@pytest.mark.parametrize("trial", range(max(6, N_TRIALS // 10)))
def test_random_dealt_sub_ensembles_equal_the_oracle(kmc, oracle, trial):
    """Dealt sub-ensembles (the multi-GPU mode without a per-half-step exchange) with random sub-ensemble counts, sizes,
    epoch lengths and run splits: per-walker results and the per-walker chain equal the oracle's restatement."""
    from kissmcmc_jl_amd.distributed import HipDealExecutor, LocalDealtEmcee
    rng = np.random.default_rng(BASE + 55000 + trial)
    P = int(rng.choice([1, 2, 4, 8]))
    nd = int(rng.choice([1, 2, 5, 8, 32, 33, 64]))
    S = int(rng.choice([2, 4, 16, 33, 128])) * 2 * P                    # S % (2 P) == 0
    while S < nd + 2:
        S += 2 * P
    G = int(rng.integers(2, 80)); nburn = int(rng.integers(0, G)); nthin = int(rng.choice([1, 1, 2, 3])); E = int(rng.integers(1, 40))
    seed = int(rng.integers(1, 2 ** 40))
    N = P * S
    th = rng.standard_normal((N, nd))
    exs = [HipDealExecutor(kmc.GaussianIso(), S, nd, G, nburn, nthin, 2.0, seed, rank=r, world=P, device=0, store_chain=True, store_logp=True) for r in range(P)]
    drv = LocalDealtEmcee(exs, N, nd, E)
    try:
        drv.set_positions(th)
        left = G
        while left > 0:
            n = int(min(left, rng.choice([1, 7, 64, 100])))
            drv.run(n)
            left -= n
        drv.sync()
        res = drv.results()
        thetas, logd = drv.gather_chain()
    finally:
        drv.close()
    label = f"trial {trial}: {P} x {S} x {nd}, G={G} nburn={nburn} nthin={nthin} E={E}"
    ref = oracle.emcee_dealt(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], N, nd, G, nburn, nthin, 2.0, seed, nthreads=4), P, E, th, store_chain=True)
    assert ref["status"] == 0, label
    np.testing.assert_array_equal(res["naccept"], ref["naccept"], err_msg=label)
    np.testing.assert_array_equal(res["positions"], ref["final_pos"], err_msg=label)
    np.testing.assert_array_equal(thetas, ref["chain"].transpose(1, 0, 2), err_msg=label)
    assert res["n"] == ref["nmoment"], label
    np.testing.assert_allclose(res["sum"], ref["sum"], rtol=1e-11, atol=1e-8, err_msg=label)


@pytest.mark.parametrize("trial", range(max(6, N_TRIALS // 15)))
def test_random_runtime_compiled_density_equals_the_oracle(kmc, oracle, monkeypatch, trial, kmc_debug):
    """Runtime-compiled densities restating a menu density -- term / pair expressions and function bodies recognised as sums over
    elements (lane-striped kernels), general bodies (rows lane-striped, evaluated per walker: `no-body-routing`), and the staged /
    generic one-walker-per-lane kernels (`no-body-vec`) -- random shapes and launch modes: identical counters and positions,
    log-pdfs to 1e-12."""
    rng = np.random.default_rng(BASE + 88000 + trial)
    form = str(rng.choice(["expr", "body", "body-general", "body-lane", "body-two-sums"]))
    if form in ("body-general", "body-lane"):
        kmc_debug.set("no-body-routing")
    if form == "body-lane":
        kmc_debug.set("no-body-vec")
    two_sums = form == "body-two-sums"          # a second accumulator that does not change the value: SepDensityN against the oracle's density
    form_label, form = form, ("expr" if form == "expr" else "body")
    name = str(rng.choice(["gauss", "rosen"]))
    nd = int(rng.choice([2, 3, 8, 17, 32, 64, 65, 130]) if name == "rosen" else rng.choice([1, 2, 5, 8, 16, 31, 32, 33, 64, 100]))
    nw = int(rng.choice([nd + 2 + nd % 2, 64, 130, 256, 1000, 2050]))
    nw = max(nw, nd + 2 + nd % 2); nw += nw % 2
    G = int(rng.integers(3, 100)); nburn = int(rng.integers(0, G)); nthin = int(rng.choice([1, 2, 3])); seed = int(rng.integers(1, 2 ** 40))
    launch = str(rng.choice(["", "graph", "eager", "updated"]))
    if launch:
        monkeypatch.setenv("KMC_LAUNCH", launch)
    if rng.random() < 0.5:
        kmcenv.no_resident(monkeypatch)
    if two_sums and name == "gauss":
        did, params = oracle.GAUSSIAN_ISO, [0.3, 1.5]
        pdf = kmc.CDensity("double s = 0.0, u = 0.0; for (int i = 0; i < n; ++i) { double t = (x[i] - p[0]) * p[1]; s += t * t; u += x[i]; } return -0.5 * s + 0.0 * u;", params=[0.3, 1.0 / 1.5])
        assert pdf.separable
        th = 0.3 + rng.standard_normal((nw, nd))
    elif two_sums:
        did, params = oracle.ROSENBROCK, [1.0, 100.0, 20.0]
        pdf = kmc.CDensity("double s = 0.0; double u = 0.0; for (int i = 0; i + 1 < n; ++i) { double d = x[i + 1] - x[i] * x[i]; double e = p[0] - x[i]; s += p[1] * (d * d) + e * e; u += x[i + 1]; } "
                           "return -(s * (1.0 / p[2])) + 0.0 * u;", params=params)
        assert pdf.separable
        th = 0.1 * rng.standard_normal((nw, nd))
    elif name == "gauss":
        did, params = oracle.GAUSSIAN_ISO, [0.3, 1.5]
        pdf = (kmc.ExprDensity("-0.5*((x-p[0])*p[1])*((x-p[0])*p[1])", params=[0.3, 1.0 / 1.5]) if form == "expr" else
               kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) { double t = (x[i] - p[0]) * p[1]; s += t * t; } return -0.5 * s;", params=[0.3, 1.0 / 1.5]))
        th = 0.3 + rng.standard_normal((nw, nd))
    else:
        did, params = oracle.ROSENBROCK, [1.0, 100.0, 20.0]
        pdf = (kmc.ExprDensity("d < n-1 ? -((p[0]-x)*(p[0]-x))/p[2] : 0.0", "-(p[1]*((y-x*x)*(y-x*x)))/p[2]", params) if form == "expr" else
               kmc.CDensity("double s = 0.0; for (int i = 0; i + 1 < n; ++i) { double d = x[i + 1] - x[i] * x[i]; double e = p[0] - x[i]; s += p[1] * (d * d) + e * e; } "
                            "return -(s * (1.0 / p[2]));", params=params))
        th = 0.1 * rng.standard_normal((nw, nd))
    label = f"trial {trial}: {form_label} {name} {nw}x{nd} G={G} nburn={nburn} nthin={nthin} launch={launch or 'auto'}"
    ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, nburn, nthin, 2.0, seed), th)
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, moments=True) as s:
        s.set_positions(th)
        s.run(G // 2); s.run(G - G // 2)
        s.sync()
        np.testing.assert_array_equal(s.naccept(), ref["naccept"], err_msg=label)
        np.testing.assert_array_equal(s.positions(), ref["final_pos"], err_msg=label)
        np.testing.assert_array_equal(s.chain(logp=False)[0], ref["chain"], err_msg=label)
        assert np.all(np.abs(s.logp() - ref["final_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"]))), label
        assert s.moments()[2] == ref["nmoment"], label


@pytest.mark.parametrize("trial", range(max(6, N_TRIALS // 15)))
def test_random_blob_density_equals_the_oracle_and_its_own_chain(kmc, oracle, monkeypatch, trial):
    """Body densities WITH blobs (reference hasblob=true on the device; src/samplers.jl:194-196, :264, :270), random shapes, blob
    widths (staged kernel up to 8 doubles, generic beyond) and launch modes: the sampler equals the oracle's run of the same
    density, and every stored blob equals the blob function of the stored position it belongs to."""
    rng = np.random.default_rng(BASE + 99000 + trial)
    nd = int(rng.choice([1, 2, 5, 8, 16, 31, 32, 33, 64, 65, 100]))
    nb = int(rng.choice([1, 2, 3, 8, 9, 20]))
    nw = int(rng.choice([nd + 2 + nd % 2, 64, 130, 256, 1000, 2050]))
    nw = max(nw, nd + 2 + nd % 2); nw += nw % 2
    G = int(rng.integers(3, 100)); nburn = int(rng.integers(0, G)); nthin = int(rng.choice([1, 2, 3])); seed = int(rng.integers(1, 2 ** 40))
    launch = str(rng.choice(["", "graph", "eager", "updated"]))
    if launch:
        monkeypatch.setenv("KMC_LAUNCH", launch)
    # blob[i] = x[i mod n] * (i + 1) for i < nb - 1 (exact in double), blob[nb - 1] = the log-density
    body = ("double s = 0.0; for (int i = 0; i < n; ++i) { double t = (x[i] - p[0]) * p[1]; s += t * t; } "
            f"for (int i = 0; i < {nb - 1}; ++i) blob[i] = x[i % n] * (double)(i + 1); blob[{nb - 1}] = -0.5 * s; return -0.5 * s;")
    pdf = kmc.CDensity(body, params=[0.3, 1.0 / 1.5], nblob=nb)
    th = 0.3 + rng.standard_normal((nw, nd))
    label = f"trial {trial}: blob density {nw}x{nd} nblob={nb} G={G} nburn={nburn} nthin={nthin} launch={launch or 'auto'}"
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.3, 1.5], nw, nd, G, nburn, nthin, 2.0, seed), th)
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, store_blobs=True) as s:
        s.set_positions(th)
        s.run(G // 3); s.sync(); s.run(G - G // 3)
        s.sync()
        np.testing.assert_array_equal(s.naccept(), ref["naccept"], err_msg=label)
        np.testing.assert_array_equal(s.positions(), ref["final_pos"], err_msg=label)
        chain, clogp = s.chain()
        blobs = s.blobs(by_walker=False)
        cur, pos, lp = s.current_blobs(), s.positions(), s.logp()
    np.testing.assert_array_equal(chain, ref["chain"], err_msg=label)
    assert blobs.shape == (chain.shape[0], nw, nb), label
    for i in range(nb - 1):
        np.testing.assert_array_equal(blobs[:, :, i], chain[:, :, i % nd] * float(i + 1), err_msg=label)
        np.testing.assert_array_equal(cur[:, i], pos[:, i % nd] * float(i + 1), err_msg=label)
    np.testing.assert_array_equal(blobs[:, :, nb - 1], clogp, err_msg=label)
    np.testing.assert_array_equal(cur[:, nb - 1], lp, err_msg=label)
