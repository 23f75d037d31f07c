"""GPU: blobs of DEVICE densities -- the reference's `pdf(theta) -> (p, blob)` under hasblob=true (src/samplers.jl:150-151,
:194-196, :208-210, :238, :264, :270) as a fixed-width side array: a CDensity body that also fills blob[0..m), carried by the
kernels next to the walker's log-pdf (no host round trip per half-step).  The acceptance cases are the reference's own blob
tests (test/runtests.jl:80-107 through test/emcee.jl:21-45) plus an exact identity: a blob that is a function of the position
must equal that function of the stored position, sample by sample."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# log-pdf of N(0, 1) per dimension; blob = {x[0], x[n-1], x[0] * x[1], the log-pdf itself}
BODY = ("double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; "
        "blob[0] = x[0]; blob[1] = x[n - 1]; blob[2] = x[0] * x[1]; blob[3] = -0.5 * s; return -0.5 * s;")


@pytest.mark.parametrize("kernel", ["default", "one-walker-per-lane"])
@pytest.mark.parametrize("nw,nd,nthin", [(24, 4, 3), (4096, 32, 1), (2048, 70, 2), (640, 7, 1)])
def test_device_blobs_follow_the_walkers_exactly(kmc, oracle, nw, nd, nthin, kernel, kmc_debug):
    """blob0s[nc] = blob1 exactly when theta0s[nc] = theta1 (:261-264), reduce_blob! exactly when stored (:268-271): the stored
    blobs equal the blob function of the stored thetas, entry by entry; and carrying blobs does not change the sampler
    (same chain as the oracle's run of the same density).  Resident kernel (24, 640 walkers), the vector kernel with the body
    evaluated per walker (default beyond), and with `no-body-vec` the staged (ndim <= 64) and generic (70) one-walker-per-lane kernels."""
    if kernel == "one-walker-per-lane":
        kmc_debug.set("no-body-vec")
    G, nburn, seed = 60, 13, 4
    th = np.random.default_rng(2).standard_normal((nw, nd))
    pdf = kmc.CDensity(BODY, nblob=4)
    assert pdf.nblob == 4
    with kmc.Sampler(pdf, nw, nd, G, nburn, nthin, 2.0, seed, store_chain=True, store_logp=True, store_blobs=True) as s:
        s.set_positions(th)
        b0 = s.current_blobs()
        np.testing.assert_array_equal(b0[:, 0], th[:, 0])
        np.testing.assert_array_equal(b0[:, 2], th[:, 0] * th[:, 1])
        s.run(G)
        s.sync()
        chain, clogp = s.chain(by_walker=True)
        blobs = s.blobs(by_walker=True)
        blobs_sm = s.blobs(by_walker=False)
        cur, pos, lp = s.current_blobs(), s.positions(), s.logp()
        how = s.describe()
        assert "blob of 4 doubles" in how
        if nw > 1024:
            assert ("half_step_vec" in how and "evaluated per walker" in how) == (kernel == "default"), how
    ns = (G - nburn) // nthin
    assert blobs.shape == (nw, ns, 4) and blobs_sm.shape == (ns, nw, 4)
    np.testing.assert_array_equal(blobs_sm.transpose(1, 0, 2), blobs)
    np.testing.assert_array_equal(blobs[:, :, 0], chain[:, :, 0])
    np.testing.assert_array_equal(blobs[:, :, 1], chain[:, :, nd - 1])
    np.testing.assert_array_equal(blobs[:, :, 2], chain[:, :, 0] * chain[:, :, 1])
    np.testing.assert_array_equal(blobs[:, :, 3], clogp)                      # the blob was computed with THIS log-pdf
    np.testing.assert_array_equal(cur[:, 0], pos[:, 0])
    np.testing.assert_array_equal(cur[:, 3], lp)
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, nthin, 2.0, seed), th)
    np.testing.assert_array_equal(chain, ref["chain"].transpose(1, 0, 2))


def test_reference_blob_cases_on_the_device(kmc):
    """reference test/runtests.jl:80-107 through test/emcee.jl:21-45: pdf = x -> (-(x+5)^2/(2*3^2), ones(1000)), 100 walkers,
    niter = 10^4; default reductions (blob_truths = 5000 x ones(1000)) and the sum reduction (blob_truths = [5000])."""
    pdf = kmc.CDensity("for (int i = 0; i < 1000; ++i) blob[i] = 1.0; const double t = x[0] + 5.0; return -(t * t) / (2.0 * 3.0 * 3.0);", nblob=1000)
    theta0s = kmc.make_theta0s(-4.0, 0.1, pdf, 100, hasblob=True, rng=5)
    samples = kmc.emcee(pdf, theta0s, niter=10 ** 4, hasblob=True, use_progress_meter=False, seed=8)
    assert [len(x) for x in samples[:3]] == [100, 100, 100] and len(samples[3]) == 100     # test/emcee.jl:29-31
    assert samples[0].shape[1] == 10 ** 4 // 100 // 2                                       # :35
    thetas, ar, logd, blobs = kmc.squash_walkers(*samples, verbose=False)
    assert len(thetas) == 10 ** 4 // 2 and len(logd) == 10 ** 4 // 2 and ar > 0.1          # :41-43
    assert len(blobs) == 10 ** 4 // 2 and all(np.array_equal(b, np.ones(1000)) for b in blobs)   # blob_truths, default reductions
    from refcases import check_mean_std
    check_mean_std(thetas, dict(name="blob case", mean=-5.0, median=-5.0, std=3.0, skew=None, tol=0.3))   # test_mean_std, runtests.jl:36-43

    def add(blobs, blob):
        blobs[0] += blob[0]

    samples = kmc.emcee(pdf, theta0s, niter=10 ** 4, hasblob=True, use_progress_meter=False, seed=9,
                        init_blobs=lambda blob0, nsamples: [0], reduce_blob=add)
    assert len(samples[3]) == 100
    thetas, ar, logd, blobs = kmc.squash_walkers(*samples, verbose=False, merge_blobs=add)
    assert blobs == [10 ** 4 // 2]                                                          # blob_truths = [10^4 / 2]


def test_device_blobs_through_the_c_abi_one_shot(kmc):
    """kmc_emcee_run with kmc_outputs.blobs: the one-shot call the Julia shim makes (KMC_STORE_BLOBS switched on by the pointer)."""
    import ctypes as C
    from kissmcmc_jl_amd import _lib
    nw, nd, G, nburn = 256, 6, 40, 10
    th = np.ascontiguousarray(np.random.default_rng(1).standard_normal((nw, nd)))
    pdf = kmc.CDensity(BODY, nblob=4)
    cfg = _lib.Config()
    cfg.dtype, cfg.density = _lib.F64, pdf.density_id
    cfg.nwalkers, cfg.ndim, cfg.ngenerations, cfg.nburnin, cfg.nthin, cfg.a_scale, cfg.seed = nw, nd, G, nburn, 1, 2.0, 3
    cfg.flags = _lib.CHAIN_BY_WALKER
    cfg.user_density = pdf.user_handle
    ns = G - nburn
    chain, blobs, acc = np.empty((nw, ns, nd)), np.empty((nw, ns, 4)), np.empty(nw)
    out = _lib.Outputs()
    dp = C.POINTER(C.c_double)
    out.chain, out.blobs, out.accept_ratio = chain.ctypes.data_as(dp), blobs.ctypes.data_as(dp), acc.ctypes.data_as(dp)
    _lib.check(_lib.lib().kmc_emcee_run(C.byref(cfg), th.ctypes.data_as(dp), C.byref(out)))
    assert out.nsamples == ns
    np.testing.assert_array_equal(blobs[:, :, 0], chain[:, :, 0])
    np.testing.assert_array_equal(blobs[:, :, 2], chain[:, :, 0] * chain[:, :, 1])
    lp, bl = pdf.eval_with_blobs(th)                                    # kmc_logpdf_blob_eval_host: pdf.(theta0s), :209-210
    np.testing.assert_array_equal(bl[:, 0], th[:, 0])
    np.testing.assert_allclose(lp, -0.5 * (th * th).sum(axis=1), rtol=1e-14)
    np.testing.assert_array_equal(bl[:, 3], lp)


def test_device_blobs_resume_init_ball_and_refusals(kmc):
    nw, nd = 512, 8
    th = np.random.default_rng(7).standard_normal((nw, nd))
    pdf = kmc.CDensity(BODY, nblob=4)
    # checkpoint / resume: the blobs of the restored positions are evaluated again
    with kmc.Sampler(pdf, nw, nd, 100, 0, 1, 2.0, 5, store_blobs=False) as s:
        s.set_positions(th)
        s.run(30)
        s.sync()
        st, cur30 = s.state(), s.current_blobs()
        s.run(20)
        s.sync()
        pos50, cur50 = s.positions(), s.current_blobs()
    with kmc.Sampler(pdf, nw, nd, 100, 0, 1, 2.0, 5) as s:
        s.restore(st)
        np.testing.assert_array_equal(s.current_blobs(), cur30)
        s.run(20)
        s.sync()
        np.testing.assert_array_equal(s.positions(), pos50)
        np.testing.assert_array_equal(s.current_blobs(), cur50)
    # device-side initial ball: the admitted points' blobs
    with kmc.Sampler(pdf, nw, nd, 100, 0, 1, 2.0, 5) as s:
        s.init_ball(np.zeros(nd), 0.1, seed=3)
        np.testing.assert_array_equal(s.current_blobs()[:, 0], s.positions()[:, 0])
    # what blobs do not combine with says so
    for kw in (dict(dtype="f32"), dict(island_gens=8, island_size=64), dict(p2p=True, shard_count=2), dict(store_chain=True, stream_chain=True),
               dict(deal_count=2)):
        with pytest.raises(kmc.KmcError):
            kmc.Sampler(pdf, nw, nd, 100, 0, 1, 2.0, 5, **kw)
    with pytest.raises(kmc.KmcError, match="KMC_STORE_BLOBS needs a body density with blobs"):
        kmc.Sampler(kmc.GaussianIso(), nw, nd, 100, 0, 1, 2.0, 5, store_blobs=True)
    with pytest.raises(NotImplementedError, match="returns a blob"):
        kmc.emcee(kmc.GaussianIso(), th, niter=nw * 10, hasblob=True, use_progress_meter=False)
    with pytest.raises(kmc.KmcError, match="nblob must be in 1"):
        kmc.CDensity("return 0.0;", nblob=5000)


def test_blob_call_sequence_in_plain_c(tmp_path):
    """examples/blob_call.c: the reference's blob case through the C ABI alone (kmc_user_density_create_body_blob,
    kmc_logpdf_blob_eval_host, kmc_outputs.blobs) -- compiled with gcc, no Python or torch in the process."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "blob_call")
    libdir = os.path.join(root, "kissmcmc.jl_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "blob_call.c"),
                           "-o", exe, "-L", libdir, "-lkissmcmc_hip", "-lm", f"-Wl,-rpath,{libdir}"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, KMC_CACHE_DIR=str(tmp_path / "cache")))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("samples 5000 blobs-that-do-not-follow-their-walker 0 ")


def test_metropolis_device_blobs(kmc, oracle):
    """The reference's `metropolis` carries blobs too (src/samplers.jl:70-72, :103, :117; the same blob cases run through
    test/metro.jl).  Many chains at once with a CDensity(nblob=m): blob0 follows each chain on the device; stored blobs equal the
    blob function of the stored states; the chains themselves equal the oracle's restatement of the same density."""
    nc, nd, niter, nburn, nthin, seed = 700, 5, 90, 20, 3, 17
    th = np.random.default_rng(4).standard_normal((nc, nd))
    pdf = kmc.CDensity(BODY, nblob=4)
    thetas, acc, logd, blobs = kmc.metropolis_chains(pdf, kmc.GaussianStep(0.7), th, niter=niter, nburnin=nburn, nthin=nthin, hasblob=True, seed=seed)
    ns = (niter - nburn) // nthin
    assert thetas.shape == (nc, ns, nd) and blobs.shape == (nc, ns, 4)
    np.testing.assert_array_equal(blobs[:, :, 0], thetas[:, :, 0])
    np.testing.assert_array_equal(blobs[:, :, 1], thetas[:, :, nd - 1])
    np.testing.assert_array_equal(blobs[:, :, 2], thetas[:, :, 0] * thetas[:, :, 1])
    np.testing.assert_array_equal(blobs[:, :, 3], logd)
    ref = oracle.metropolis(oracle.GAUSSIAN_ISO, [0.0, 1.0], th, 0.7, niter, nburn, nthin, seed)
    np.testing.assert_array_equal(np.rint(acc * (niter - nburn)).astype(np.int64), ref["naccept"])
    np.testing.assert_allclose(thetas.transpose(1, 0, 2), ref["chain"], rtol=1e-11, atol=1e-11)
    # the caller's reduction, fed each chain's series in order: the reference's sum case (test/runtests.jl:94-107), here on blob[2]
    def add(bs, b):
        bs[0] += b[2]
    _, _, _, sums = kmc.metropolis_chains(pdf, kmc.GaussianStep(0.7), th, niter=niter, nburnin=nburn, nthin=nthin, hasblob=True, seed=seed,
                                          init_blobs=lambda blob0, n: [0.0], reduce_blob=add)
    np.testing.assert_allclose([s[0] for s in sums], (thetas[:, :, 0] * thetas[:, :, 1]).sum(axis=1), rtol=1e-12, atol=1e-12)
    # one chain, the reference's signature and its blob case: pdf = x -> (-(x+5)^2/18, ones(...)); blob_truths = nsamples x ones
    one = kmc.CDensity("for (int i = 0; i < 16; ++i) blob[i] = 1.0; const double t = x[0] + 5.0; return -(t * t) / 18.0;", nblob=16)
    t1, a1, l1, b1 = kmc.metropolis(one, kmc.GaussianStep(9.0), -4.0, niter=10 ** 4, hasblob=True, use_progress_meter=False, seed=3)
    assert len(t1) == 5000 and len(b1) == 5000 and np.array_equal(b1, np.ones((5000, 16))) and 0.15 < a1 < 0.45   # test/metro.jl:14-17
    with pytest.raises(NotImplementedError, match="GaussianStep"):
        kmc.metropolis_chains(pdf, lambda t: t + 0.1, th, niter=10, hasblob=True, seed=1)
