"""GPU: dealt sub-ensembles (kmc_config.deal_count, distributed.DealtEmcee) -- each sub-ensemble an ordinary HIP sampler,
walkers re-dealt between epochs -- against the oracle's restatement kmco_emcee_dealt: positions and acceptance counters
bit-identical per WALKER, log-pdfs within 1e-12, moments within 1e-11.  P sub-ensembles in one process on one GPU
(copies instead of the collective), and 2 processes sharing the GPU with the real driver over gloo (staged through the
host; on a multi-GPU node the same driver runs RCCL all_to_all_single on device buffers)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(res, ref):
    assert ref["status"] == 0
    np.testing.assert_array_equal(res["naccept"], ref["naccept"])
    np.testing.assert_array_equal(res["positions"], ref["final_pos"])
    assert np.all(np.abs(res["logp"] - ref["final_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"])))
    assert res["n"] == ref["nmoment"]
    np.testing.assert_allclose(res["sum"], ref["sum"], rtol=1e-11, atol=1e-8)
    np.testing.assert_allclose(res["sumsq"], ref["sumsq"], rtol=1e-11, atol=1e-8)


CASES = {
    # name: (density ctor, oracle id, params, P, S, ndim, G, nburn, nthin, E, scale)
    "gauss_4x1024x32_eager_epochs": ("GaussianIso", 0, [0.0, 1.0], 4, 1024, 32, 70, 20, 1, 16, 1.0),
    "gauss_2x4096x32_graph_epochs": ("GaussianIso", 0, [0.0, 1.0], 2, 4096, 32, 200, 60, 3, 64, 1.0),
    "rosen_2x1024x64_draw_ring": ("Rosenbrock", 2, [1.0, 100.0, 20.0], 2, 1024, 64, 23, 5, 1, 5, 0.1),
    "gauss_2x512x5_ragged": ("GaussianIso", 0, [0.0, 1.0], 2, 512, 5, 30, 10, 2, 4, 1.0),
    "gauss_2x1024x256_long_rows": ("GaussianIso", 0, [0.0, 1.0], 2, 1024, 256, 12, 3, 1, 4, 1.0),
    "expo_2x128x4_resident": ("Exponential", 1, [1.0], 2, 128, 4, 90, 30, 1, 20, None),
    "gauss_8x2048x32": ("GaussianIso", 0, [0.0, 1.0], 8, 2048, 32, 40, 8, 1, 8, 1.0),
}


@pytest.mark.parametrize("case", list(CASES))
def test_logical_sub_ensembles_on_one_gpu_equal_the_oracle(kmc, oracle, case):
    from kissmcmc_jl_amd.distributed import HipDealExecutor, LocalDealtEmcee
    name, did, params, P, S, nd, G, nburn, nthin, E, scale = CASES[case]
    pdf = getattr(kmc, name)()
    N = P * S
    rng = np.random.default_rng(3)
    th = 0.5 + 0.1 * np.abs(rng.standard_normal((N, nd))) if scale is None else scale * rng.standard_normal((N, nd))
    exs = [HipDealExecutor(pdf, S, nd, G, nburn, nthin, 2.0, 4242, rank=r, world=P, device=0, store_chain=True, store_logp=True) for r in range(P)]
    drv = LocalDealtEmcee(exs, N, nd, E)
    try:
        drv.set_positions(th)
        drv.run(G // 2)
        drv.run(G - G // 2)
        drv.sync()
        res = drv.results()
        thetas, logd = drv.gather_chain()               # stored by slot on the device, re-filed by walker through every deal
        assert "dealt sub-ensemble" in exs[0].sampler.describe()
    finally:
        drv.close()
    ref = oracle.emcee_dealt(oracle.make_config(did, params, N, nd, G, nburn, nthin, 2.0, 4242, nthreads=8), P, E, th, store_chain=True)
    _check(res, ref)
    np.testing.assert_array_equal(thetas, ref["chain"].transpose(1, 0, 2))
    assert np.all(np.abs(logd - ref["chain_logp"].T) <= 1e-12 * np.maximum(1.0, np.abs(ref["chain_logp"].T)))


def test_deal_pack_layout_and_walker_ids(kmc, oracle):
    """One pack straight through the C ABI: row t = (A j + C) mod S of the send buffer holds slot j's
    {position, log-pdf, (id << 32 | naccept)} (include/kissmcmc_hip.h)."""
    import torch
    S, nd, rank, P, seed = 256, 6, 3, 4, 77
    th = np.random.default_rng(0).standard_normal((S, nd))
    with kmc.Sampler(kmc.GaussianIso(), S, nd, 20, 0, 1, 2.0, seed, moments=True, deal_rank=rank, deal_count=P) as s:
        s.set_positions(th)
        np.testing.assert_array_equal(s.walker_ids(), rank * S + np.arange(S))
        s.run(9)
        s.sync()
        pos, logp, nacc = s.positions(), s.logp(), s.naccept()
        buf = torch.zeros((S, nd + 2), dtype=torch.float64, device="cuda")
        s.deal_pack(5, buf.data_ptr())
        s.sync()
        b = buf.cpu().numpy()
        a, c = oracle.deal_perm(seed, 5, rank, S)
        t = (a * np.arange(S) + c) % S
        np.testing.assert_array_equal(b[t, :nd], pos)
        np.testing.assert_array_equal(b[t, nd], logp)
        w = np.ascontiguousarray(b[t, nd + 1]).view(np.uint64)
        np.testing.assert_array_equal((w & np.uint64(0xFFFFFFFF)).astype(np.int64), nacc)
        np.testing.assert_array_equal((w >> np.uint64(32)).astype(np.int64), rank * S + np.arange(S))
        s.deal_unpack(buf.data_ptr())                       # takes the shuffled rows in order
        s.sync()
        np.testing.assert_array_equal(s.positions(), b[:, :nd])
        inv = np.empty(S, dtype=np.int64)
        inv[t] = np.arange(S)
        np.testing.assert_array_equal(s.walker_ids(), rank * S + inv)
    # the sub-ensemble's stream is the reference's algorithm under its own key: equal to the oracle with that key
    cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], S, nd, 9, 0, 1, 2.0, oracle.deal_seed(seed, rank))
    ref = oracle.emcee(cfg, th, store_chain=False)
    np.testing.assert_array_equal(pos, ref["final_pos"])
    np.testing.assert_array_equal(nacc, ref["naccept"])


def test_dealt_init_ball_draws_rows_of_one_global_ball(kmc, oracle):
    S, nd, P = 512, 3, 4
    ref = oracle.init_ball(oracle.EXPONENTIAL, [1.0], 0.02, 0.1, P * S, nd, seed=9)
    for r in (0, 2):
        with kmc.Sampler(kmc.Exponential(), S, nd, 10, deal_rank=r, deal_count=P) as s:
            s.init_ball(0.02, 0.1, seed=9)
            np.testing.assert_allclose(s.positions(), ref["pos"][r * S:(r + 1) * S], rtol=1e-11, atol=1e-13)
            np.testing.assert_array_equal(s.walker_ids(), r * S + np.arange(S))


def test_dealt_config_validation(kmc):
    with pytest.raises(kmc.KmcError, match="divisible by deal_count"):
        kmc.Sampler(kmc.GaussianIso(), 100, 2, 10, deal_rank=0, deal_count=3)
    with pytest.raises(kmc.KmcError, match="dealt sub-ensembles"):
        kmc.Sampler(kmc.GaussianIso(), 128, 2, 10, deal_rank=0, deal_count=2, store_chain=True, stream_chain=True)
    with pytest.raises(kmc.KmcError, match="deal_rank"):
        kmc.Sampler(kmc.GaussianIso(), 128, 2, 10, deal_rank=2, deal_count=2)
    with kmc.Sampler(kmc.GaussianIso(), 128, 2, 10) as s:
        import torch
        buf = torch.zeros((128, 4), dtype=torch.float64, device="cuda")
        with pytest.raises(kmc.KmcError, match="deal_count"):
            s.deal_pack(0, buf.data_ptr())


NW2, ND2, G2, NB2, E2, SEED2 = 8192, 32, 150, 40, 64, 31


def _theta2():
    return np.random.default_rng(12).standard_normal((NW2, ND2))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import kissmcmc_jl_amd as kmc
    from kissmcmc_jl_amd.distributed import DealtEmcee, HipDealExecutor
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = HipDealExecutor(kmc.GaussianIso(), NW2 // world, ND2, G2, NB2, 10, 2.0, SEED2, rank=rank, world=world, device=0,
                             store_chain=True, store_logp=True)
        drv = DealtEmcee(ex, NW2, ND2, E2)
        drv.set_positions(_theta2())
        drv.run(G2)
        drv.sync()
        res = drv.results()
        res["thetas"], res["logd"] = drv.gather_chain()
        local, _, walker = drv.chain()
        assert local.shape == (11, NW2 // world, ND2) and walker.shape == (11, NW2 // world)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), **res)
        drv.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    from portpick import rendezvous_port
    return rendezvous_port()


def test_two_processes_sharing_the_gpu_equal_the_oracle(oracle, tmp_path):
    """The real driver (epochs of hipGraph replays, pack, all_to_all_single, unpack), one process per sub-ensemble."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = oracle.emcee_dealt(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW2, ND2, G2, NB2, 10, 2.0, SEED2, nthreads=8), world, E2, _theta2(),
                             store_chain=True)
    for r in range(world):
        z = dict(np.load(os.path.join(str(tmp_path), f"r{r}.npz")))
        z["n"] = int(z["n"])
        _check(z, ref)
        np.testing.assert_array_equal(z["thetas"], ref["chain"].transpose(1, 0, 2))       # 11 samples per walker, through 2 deals


def test_dealt_driver_world1_through_rccl(kmc, oracle):
    """World size 1 on the nccl (= RCCL) backend with the collective forced: all_to_all_single on the device buffers,
    issued inside the executor's stream -- the call sequence of a multi-GPU run with its stream ordering."""
    import torch
    import torch.distributed as dist
    from kissmcmc_jl_amd.distributed import DealtEmcee, HipDealExecutor
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        nw, nd, G, nburn, E, seed = 4096, 32, 200, 40, 64, 3
        th = np.random.default_rng(6).standard_normal((nw, nd))
        ex = HipDealExecutor(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, seed, rank=0, world=1, device=0)
        drv = DealtEmcee(ex, nw, nd, E, always_collective=True)
        drv.set_positions(th)
        drv.run(G)
        drv.sync()
        res = drv.results()
        assert drv.deals == G // E
        drv.close()
        ref = oracle.emcee_dealt(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 1, 2.0, seed, nthreads=8), 1, E, th)
        _check(res, ref)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("stop_at", [37, 48, 5])
def test_dealt_checkpoint_resume_equals_the_uninterrupted_oracle_run(kmc, oracle, stop_at):
    """state() / restore() of the sub-ensembles (the slot -> walker map is replayed from the deal permutations): stop inside
    an epoch, exactly after a deal, and before the first deal; the resumed run ends where the oracle's uninterrupted run does."""
    from kissmcmc_jl_amd.distributed import HipDealExecutor, LocalDealtEmcee
    P, S, nd, G, nburn, E, seed = 4, 512, 8, 90, 20, 16, 777
    N = P * S
    th = np.random.default_rng(6).standard_normal((N, nd))
    mk = lambda: LocalDealtEmcee([HipDealExecutor(kmc.GaussianIso(), S, nd, G, nburn, 1, 2.0, seed, rank=r, world=P, device=0) for r in range(P)], N, nd, E)
    a = mk()
    try:
        a.set_positions(th)
        a.run(stop_at)
        a.sync()
        states = a.state()
    finally:
        a.close()
    b = mk()
    try:
        b.restore(states)
        b.run(G - stop_at)
        b.sync()
        res = b.results()
    finally:
        b.close()
    ref = oracle.emcee_dealt(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], N, nd, G, nburn, 1, 2.0, seed, nthreads=8), P, E, th)
    assert ref["status"] == 0
    np.testing.assert_array_equal(res["positions"], ref["final_pos"])
    np.testing.assert_array_equal(res["naccept"], ref["naccept"])
    assert np.all(np.abs(res["logp"] - ref["final_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"])))


def test_emcee_dealt_front_end_returns_the_reference_tuple(kmc, oracle):
    """distributed.emcee_dealt (world 1 here: one sub-ensemble, re-shuffled every epoch): the reference's bookkeeping and tuple."""
    from kissmcmc_jl_amd.distributed import emcee_dealt
    nw, nd = 256, 3
    th = np.random.default_rng(2).standard_normal((nw, nd))
    thetas, acc, logd, blobs = emcee_dealt(kmc.GaussianIso(), th, niter=nw * 120, nthin=2, seed=19, epoch_gens=25)
    ref = oracle.emcee_dealt(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, 120, 60, 2, 2.0, 19), 1, 25, th, store_chain=True)
    assert blobs is None and thetas.shape == (nw, 30, nd) and logd.shape == (nw, 30)
    np.testing.assert_array_equal(thetas, ref["chain"].transpose(1, 0, 2))
    np.testing.assert_array_equal(acc, ref["accept_ratio"])
    sq = kmc.squash_walkers(thetas, acc, logd, verbose=False)
    assert sq[0].shape == (nw * 30, nd)
