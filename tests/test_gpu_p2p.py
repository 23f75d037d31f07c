"""GPU: peer-to-peer walker sharding (KMC_P2P).  One GPU validates the PROTOCOL: two (or more)
PROCESSES on the same device exchange IPC handles, read each other's rows through the peer-mapped
pointers and order their half-steps with the progress flags, as 8 processes on 8 GPUs do over xGMI;
results must equal the unsharded oracle run bit for bit.  What one GPU cannot show is cross-GPU cache
coherence (every "peer" shares one L2 and one HBM here): the pull kernels therefore read peer rows with
system-scope loads, and bench.py verifies the timed multi-GPU run against the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NW, ND, G, NBURN, SEED = 2048, 32, 150, 40, 4711


def _theta0():
    return np.random.default_rng(8).standard_normal((NW, ND))


def test_p2p_single_rank_equals_plain_sampler(kmc, oracle):
    th = _theta0()
    with kmc.Sampler(kmc.GaussianIso(), NW, ND, G, NBURN, 1, 2.0, SEED, moments=True, p2p=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        pos, nacc, mom = s.positions(), s.naccept(), s.moments()
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED), th, store_chain=False)
    np.testing.assert_array_equal(pos, ref["final_pos"])
    np.testing.assert_array_equal(nacc, ref["naccept"])
    assert mom[2] == ref["nmoment"]


def _worker(rank, world, port, outdir, plan, late_rank=-1):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import kissmcmc_jl_amd as kmc
    from kissmcmc_jl_amd.distributed import P2PEmcee
    if plan == "rosen-ragged":
        return _worker_rosen(rank, world, port, outdir)
    push = plan == "push"
    if plan and not push:
        os.environ["KMC_PLAN"] = plan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # rendezvous only (same GPU: RCCL would refuse)
    try:
        drv = P2PEmcee(kmc.GaussianIso(), NW, ND, G, NBURN, 1, 2.0, SEED, device=0, push=push)
        if push:
            assert "KMC_P2P_PUSH" in drv.sampler.describe()
        drv.set_positions(_theta0())
        if rank == late_rank:
            import time
            time.sleep(1.5)              # uneven start: the others spin on this rank's progress flag meanwhile
        drv.run(G)                       # graph replays + eager tail, all enqueued at once
        drv.sync()
        pos, logp, nacc = drv.positions(), drv.logp(), drv.naccept()
        s, q, n = drv.moments()
        if rank == 0:
            np.savez(os.path.join(outdir, "out.npz"), pos=pos, logp=logp, nacc=nacc, s=s, q=q, n=n)
        drv.close()
    finally:
        dist.destroy_process_group()


RNW, RND, RG = 768, 9, 90      # odd ndim: padded rows + masked tail chunk through the peer pointers


def _rosen_theta0():
    return 0.1 * np.random.default_rng(12).standard_normal((RNW, RND))


def _worker_rosen(rank, world, port, outdir):
    import torch.distributed as dist
    import kissmcmc_jl_amd as kmc
    from kissmcmc_jl_amd.distributed import P2PEmcee
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        drv = P2PEmcee(kmc.Rosenbrock(), RNW, RND, RG, 30, 1, 2.0, 99, device=0, store_chain=True, store_logp=True)
        drv.set_positions(_rosen_theta0())
        drv.run(RG)
        drv.sync()
        pos, nacc = drv.positions(), drv.naccept()
        thetas, logd = drv.gather_chain()                # every shard's samples by walker, in global walker order
        if rank == 0:
            np.savez(os.path.join(outdir, "rosen.npz"), pos=pos, nacc=nacc, thetas=thetas, logd=logd)
        drv.close()
        # the reference's call over the ranks: same tuple as the one-GPU emcee with this seed
        from kissmcmc_jl_amd.distributed import emcee_p2p
        t2, acc2, l2, _ = emcee_p2p(kmc.Rosenbrock(), _rosen_theta0(), niter=RNW * RG, nburnin=RNW * 30, seed=99, device=0)
        if rank == 0:
            np.savez(os.path.join(outdir, "rosen_front.npz"), thetas=t2, acc=acc2, logd=l2)
    finally:
        dist.destroy_process_group()


def test_p2p_rosenbrock_odd_ndim(oracle, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(3, _free_port(), str(tmp_path), "rosen-ragged"), nprocs=3, join=True)
    ref = oracle.emcee(oracle.make_config(oracle.ROSENBROCK, [1.0, 100.0, 20.0], RNW, RND, RG, 30, 1, 2.0, 99), _rosen_theta0())
    z = np.load(os.path.join(str(tmp_path), "rosen.npz"))
    np.testing.assert_array_equal(z["nacc"], ref["naccept"])
    np.testing.assert_array_equal(z["pos"], ref["final_pos"])
    np.testing.assert_array_equal(z["thetas"], ref["chain"].transpose(1, 0, 2))           # three shards' chains, by walker
    assert np.all(np.abs(z["logd"] - ref["chain_logp"].T) <= 1e-12 * np.maximum(1.0, np.abs(ref["chain_logp"].T)))
    f = np.load(os.path.join(str(tmp_path), "rosen_front.npz"))                            # distributed.emcee_p2p
    np.testing.assert_array_equal(f["thetas"], ref["chain"].transpose(1, 0, 2))
    np.testing.assert_array_equal(f["acc"], ref["accept_ratio"])


def _free_port():
    from portpick import rendezvous_port
    return rendezvous_port()


# ("push": accepted rows written into every peer's local copy of the shard, read there with system-scope loads -- round 5)
@pytest.mark.parametrize("world,plan", [(2, ""), (4, ""), (2, "generic"), (2, "push"), (4, "push")])   # the GPU box allows 6 processes on the card: parent + 4 ranks at most
def test_p2p_processes_sharing_one_gpu_equal_oracle(oracle, tmp_path, world, plan):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), plan), nprocs=world, join=True)
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED),
                       _theta0(), store_chain=False)
    z = np.load(os.path.join(str(tmp_path), "out.npz"))
    np.testing.assert_array_equal(z["nacc"], ref["naccept"])
    np.testing.assert_array_equal(z["pos"], ref["final_pos"])
    assert np.all(np.abs(z["logp"] - ref["final_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"])))
    assert int(z["n"]) == ref["nmoment"]
    np.testing.assert_allclose(z["s"], ref["sum"], rtol=1e-11, atol=1e-9)


def test_p2p_tolerates_a_late_rank(oracle, tmp_path):
    """One rank starts 1.5 s after the others: they wait inside their first half-step (bounded spin on
    the late rank's progress flag) and the result is still bit-exact."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "", 1), nprocs=2, join=True)
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED),
                       _theta0(), store_chain=False)
    z = np.load(os.path.join(str(tmp_path), "out.npz"))
    np.testing.assert_array_equal(z["nacc"], ref["naccept"])
    np.testing.assert_array_equal(z["pos"], ref["final_pos"])


def _long_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import kissmcmc_jl_amd as kmc
    from kissmcmc_jl_amd.distributed import P2PEmcee
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nw, nd, G = 32768, 32, 700
        th = np.random.default_rng(99).standard_normal((nw, nd))
        drv = P2PEmcee(kmc.GaussianIso(), nw, nd, G, 100, 1, 2.0, 2718, device=0)
        drv.set_positions(th)
        drv.run(G)
        drv.sync()
        pos, nacc = drv.positions(), drv.naccept()
        s, q, n = drv.moments()
        if rank == 0:
            np.savez(os.path.join(outdir, "long.npz"), pos=pos, nacc=nacc, s=s, n=n)
        drv.close()
    finally:
        dist.destroy_process_group()


def test_p2p_long_run_equals_unsharded_gpu_run(kmc, tmp_path):
    """700 generations (ten hipGraph replays + tail) of a 32 768-walker ensemble over two processes:
    1400 flag-ordered half-steps with peer reads; any stale or torn row would break bit-equality with
    the one-process run of the same ensemble."""
    import torch.multiprocessing as mp
    mp.spawn(_long_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    nw, nd, G = 32768, 32, 700
    th = np.random.default_rng(99).standard_normal((nw, nd))
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 100, 1, 2.0, 2718, moments=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        pos, nacc, mom = s.positions(), s.naccept(), s.moments()
    z = np.load(os.path.join(str(tmp_path), "long.npz"))
    np.testing.assert_array_equal(z["pos"], pos)
    np.testing.assert_array_equal(z["nacc"], nacc)
    assert int(z["n"]) == mom[2]
    np.testing.assert_allclose(z["s"], mom[0], rtol=1e-11, atol=1e-8)


def _fine_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import kissmcmc_jl_amd as kmc
    from kissmcmc_jl_amd.distributed import P2PEmcee
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        drv = P2PEmcee(kmc.GaussianIso(), NW, ND, G, NBURN, 1, 2.0, SEED, device=0, finegrained=True)
        drv.set_positions(_theta0())
        drv.run(G)
        drv.sync()
        pos, nacc = drv.positions(), drv.naccept()
        if rank == 0:
            np.savez(os.path.join(outdir, "fine.npz"), pos=pos, nacc=nacc)
        drv.close()
    finally:
        dist.destroy_process_group()


def test_p2p_with_finegrained_rows(oracle, tmp_path):
    """KMC_P2P_FINEGRAINED: rows in fine-grained device memory (what bench.py tries if the plain mapping
    fails its self-check on a multi-GPU node)."""
    import torch.multiprocessing as mp
    mp.spawn(_fine_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED),
                       _theta0(), store_chain=False)
    z = np.load(os.path.join(str(tmp_path), "fine.npz"))
    np.testing.assert_array_equal(z["nacc"], ref["naccept"])
    np.testing.assert_array_equal(z["pos"], ref["final_pos"])


@pytest.mark.parametrize("kw", [dict(), dict(p2p_push=True)], ids=["pull", "push"])
def test_two_shards_in_one_process(kmc, oracle, kw):
    """kmc_sampler_p2p_connect_local: both shards live in this process and run concurrently on their own streams,
    ordered by the same progress flags (what scripts/p2p_local_bench.py times); result = the oracle's."""
    import torch
    th = _theta0()
    G = 128         # whole hipGraph chunks only: eager launches of two streams of ONE process are not reliably concurrent
    shards = [kmc.Sampler(kmc.GaussianIso(), NW, ND, G, NBURN, 1, 2.0, SEED, moments=True, shard_rank=r, shard_count=2, p2p=True, **kw)
              for r in range(2)]
    # the shards spin on each other's progress flags, so their streams must sit on DIFFERENT hardware queues: streams of
    # one process share a small pool of queues round-robin (which two land together depends on how many streams the
    # process has created before), streams of different priority never share one
    streams = [torch.cuda.Stream(device=0, priority=-1), torch.cuda.Stream(device=0, priority=0)]
    try:
        for sh, st in zip(shards, streams):
            sh.set_stream(st.cuda_stream)
        kmc.Sampler.p2p_connect_local(shards)
        for sh in shards:
            sh.set_positions(th)
        for sh in shards:
            sh.run(G)
        for sh in shards:
            sh.sync()
        from kissmcmc_jl_amd.distributed import local_to_global
        pos = local_to_global([sh.positions() for sh in shards], NW, 2)
        nacc = local_to_global([sh.naccept() for sh in shards], NW, 2)
    finally:
        for sh in shards:
            sh.close()
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED), th, store_chain=False)
    np.testing.assert_array_equal(pos, ref["final_pos"])
    np.testing.assert_array_equal(nacc, ref["naccept"])


def test_link_probe_reads_a_peers_rows_without_changing_anything(kmc):
    """kmc_sampler_p2p_link_probe (bench.py's `link-probe` rung): whole rows of a peer's shard gathered at random row indices with the pull's
    system-scope loads + the runtime's copy of the shard -- here both "links" are this GPU's own memory.  Rates are positive and plausible,
    the probe is read-only, and it refuses a sampler that is not a connected KMC_P2P shard."""
    th = _theta0()
    shards = [kmc.Sampler(kmc.GaussianIso(), NW, ND, 64, 0, 1, 2.0, SEED, shard_rank=r, shard_count=2, p2p=True) for r in range(2)]
    try:
        with pytest.raises(kmc.KmcError, match="connected"):
            shards[0].p2p_link_probe(1, 1024)
        kmc.Sampler.p2p_connect_local(shards)
        for sh in shards:
            sh.set_positions(th)
        before = [sh.positions() for sh in shards]
        for me, peer in ((0, 1), (1, 0), (0, 0)):
            g, c = shards[me].p2p_link_probe(peer, 1 << 16, reps=5)
            assert 1.0 < g < 8000.0 and 1.0 < c < 8000.0, (me, peer, g, c)          # GB/s: above a crawl, below the HBM spec
        with pytest.raises(kmc.KmcError):
            shards[0].p2p_link_probe(2, 1024)
        for sh, b in zip(shards, before):
            np.testing.assert_array_equal(sh.positions(), b)
    finally:
        for sh in shards:
            sh.close()
    with kmc.Sampler(kmc.GaussianIso(), NW, ND, 64, 0, 1, 2.0, SEED) as plain:
        with pytest.raises(kmc.KmcError):
            plain.p2p_link_probe(0, 1024)


@pytest.mark.parametrize("form", ["expr", "body", "body-routed", "body-vec"])
def test_two_p2p_shards_with_a_runtime_compiled_density(kmc, oracle, form, kmc_debug):
    """Runtime-compiled densities run under KMC_P2P too (the pull kernels are instantiated with the user's functor):
    Rosenbrock as term / pair expressions (lane-striped kernel) and as a function body (one walker per lane),
    two shards in one process, result = the oracle's menu Rosenbrock."""
    import torch
    nw, nd, G, nburn, seed = 1024, 16, 128, 30, 99
    th = 0.1 * np.random.default_rng(3).standard_normal((nw, nd))
    if form == "body":
        kmc_debug.set("no-body-routing")          # the one-walker-per-lane pull kernel; "body-routed": the same body, recognised as a sum over elements
        kmc_debug.set("no-body-vec")
    if form == "body-vec":
        kmc_debug.set("no-body-routing")          # a general body under the pull exchange: rows lane-striped, the body evaluated per walker
    if form == "expr":
        pdf = kmc.ExprDensity("d < n-1 ? -((p[0]-x)*(p[0]-x))/p[2] : 0.0", "-(p[1]*((y-x*x)*(y-x*x)))/p[2]", [1.0, 100.0, 20.0])
    else:
        pdf = kmc.CDensity("double s = 0.0; for (int i = 0; i + 1 < n; ++i) { double d = x[i + 1] - x[i] * x[i]; double e = p[0] - x[i]; "
                           "s += p[1] * (d * d) + e * e; } return -(s * (1.0 / p[2]));", params=[1.0, 100.0, 20.0])
    shards = [kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed, moments=True, shard_rank=r, shard_count=2, p2p=True) for r in range(2)]
    streams = [torch.cuda.Stream(device=0, priority=-1), torch.cuda.Stream(device=0, priority=0)]
    try:
        assert ("half_step_generic" if form == "body" else "half_step_vec") in shards[0].describe()
        for sh, st in zip(shards, streams):
            sh.set_stream(st.cuda_stream)
        kmc.Sampler.p2p_connect_local(shards)
        for sh in shards:
            sh.set_positions(th)
        for sh in shards:
            sh.run(G)
        for sh in shards:
            sh.sync()
        from kissmcmc_jl_amd.distributed import local_to_global
        pos = local_to_global([sh.positions() for sh in shards], nw, 2)
        nacc = local_to_global([sh.naccept() for sh in shards], nw, 2)
    finally:
        for sh in shards:
            sh.close()
    ref = oracle.emcee(oracle.make_config(oracle.ROSENBROCK, [1.0, 100.0, 20.0], nw, nd, G, nburn, 1, 2.0, seed), th, store_chain=False)
    np.testing.assert_array_equal(pos, ref["final_pos"])
    np.testing.assert_array_equal(nacc, ref["naccept"])


def test_removed_exchange_variants_are_refused(kmc):
    """Lazy pull and the folded signal (rounds 1-4, behind -DKMC_P2P_EXPERIMENTAL) read peer-written memory through the local L2 or published completion from
    inside the kernel -- nothing one GPU can validate, and by DESIGN section 7's bytes per link unable to move the fabric bound: removed in round 5, their
    flag bits refused by validation with the reason.  The push of accepted rows stays: its readers use system-scope loads, like the pull's."""
    import ctypes as C
    from kissmcmc_jl_amd import _lib
    with kmc.Sampler(kmc.GaussianIso(), NW, ND, G, NBURN, 1, 2.0, SEED, shard_rank=0, shard_count=2, p2p=True, p2p_push=True) as s:
        assert "KMC_P2P_PUSH" in s.describe()
    for bit in (1 << 8, 1 << 10):
        cfg = _lib.Config(dtype=_lib.F64, density=_lib.GAUSSIAN_ISO, nwalkers=NW, ndim=ND, ngenerations=G, nburnin=NBURN, nthin=1, a_scale=2.0, seed=SEED,
                          flags=_lib.P2P | bit, device=0, shard_rank=0, shard_count=2)
        cfg.params[0], cfg.params[1] = 0.0, 1.0
        assert _lib.lib().kmc_validate(C.byref(cfg)) == _lib.ERR_UNSUPPORTED
        assert b"removed in round 5" in _lib.lib().kmc_last_error()
