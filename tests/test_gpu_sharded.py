"""GPU: walker sharding on ONE device.  P logical shards (P samplers with shard_rank r sharing one
position buffer, a device-local stand-in for the RCCL all-gather) must reproduce the unsharded
run bit for bit; and the torch.distributed driver runs with world_size 1 on the nccl backend."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,form", [(2, "menu"), (4, "menu"), (8, "menu"), (2, "body"), (4, "body")])
def test_logical_shards_on_one_gpu_equal_unsharded(kmc, oracle, P, form):
    import torch
    nw, nd, G, nburn, seed = 512, 32, 12, 4, 21
    th = np.random.default_rng(1).standard_normal((nw, nd))
    # "body": the runtime-compiled function-body form of the same density (its staged one-walker-per-lane kernel)
    pdf = kmc.GaussianIso() if form == "menu" else kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) { double t = (x[i] - p[0]) * p[1]; s += t * t; } return -0.5 * s;", params=[0.0, 1.0])
    pos = torch.empty((nw, nd), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    shards = []
    for r in range(P):
        s = kmc.Sampler(pdf, nw, nd, G, nburn, 1, 2.0, seed, moments=True, use_graph=False, shard_rank=r, shard_count=P)
        s.bind_positions(pos.data_ptr())
        s.set_stream(stream)
        s.set_positions(th)
        shards.append(s)
    for g in range(G):
        for half in (0, 1):
            for s in shards:        # same stream: all shards of a half-step, then the next half-step
                s.half_step(half)
    torch.cuda.synchronize()
    cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 1, 2.0, seed)
    ref = oracle.emcee(cfg, th, store_chain=False)
    np.testing.assert_array_equal(pos.cpu().numpy(), ref["final_pos"])
    nacc = sum(s.naccept() for s in shards)
    np.testing.assert_array_equal(nacc, ref["naccept"])
    S = sum(s.moments()[0] for s in shards); Q = sum(s.moments()[1] for s in shards); n = sum(s.moments()[2] for s in shards)
    assert n == ref["nmoment"]
    np.testing.assert_allclose(S, ref["sum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(Q, ref["sumsq"], rtol=1e-11, atol=1e-9)
    with pytest.raises(kmc.KmcError, match="shard_count == 1"):
        shards[0].run(1)
    for s in shards:
        s.close()


def test_distributed_driver_world1_nccl(kmc, oracle):
    import torch
    import torch.distributed as dist
    from kissmcmc_jl_amd.distributed import HipShardExecutor, ShardedEmcee
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        nw, nd, G, nburn, seed = 256, 32, 10, 3, 8
        th = np.random.default_rng(2).standard_normal((nw, nd))
        ex = HipShardExecutor(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, seed, rank=0, world=1)
        ex.set_positions(th)
        drv = ShardedEmcee(ex, nw, nd)
        drv.run(G)
        ex.sync()
        cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 1, 2.0, seed)
        ref = oracle.emcee(cfg, th, store_chain=False)
        np.testing.assert_array_equal(drv.positions(), ref["final_pos"])
        np.testing.assert_array_equal(drv.naccept(), ref["naccept"])
        s, q, n = drv.moments()
        assert n == ref["nmoment"]
        np.testing.assert_allclose(s, ref["sum"], rtol=1e-11, atol=1e-9)
        ex.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [True, False])
def test_native_rccl_allgather_world1_equals_oracle(kmc, oracle, use_graph):
    """AllGatherEmcee: kernel + in-place ncclAllGather per half-step, enqueued by kmc_sampler_run (one rank: the RCCL calls,
    their capture into the hipGraph chunks and the eager form are real; the partition itself is covered by the logical-shard
    and gloo tests)."""
    from kissmcmc_jl_amd.distributed import AllGatherEmcee
    nw, nd, G, nburn, seed = 4096, 32, 200, 50, 12
    th = np.random.default_rng(4).standard_normal((nw, nd))
    drv = AllGatherEmcee(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, seed, device=0, use_graph=use_graph)
    try:
        drv.set_positions(th)
        drv.run(G)
        drv.sync()
        how = drv.sampler.describe()
        assert "RCCL all-gather" in how
        pos, nacc, (s, q, n) = drv.positions(), drv.naccept(), drv.moments()
    finally:
        drv.close()
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 1, 2.0, seed, nthreads=8), th, store_chain=False)
    np.testing.assert_array_equal(pos, ref["final_pos"])
    np.testing.assert_array_equal(nacc, ref["naccept"])
    assert n == ref["nmoment"]
    np.testing.assert_allclose(s, ref["sum"], rtol=1e-11, atol=1e-9)
    print(how)


def test_rccl_capture_vote_overrides_this_ranks_own_capture(kmc, oracle, monkeypatch):
    """kmc_sampler_rccl_capture / _set_capture: a rank whose own capture succeeded still launches one by one when the ranks'
    vote (MIN over the capture outcomes, distributed.AllGatherEmcee) says so -- never a mixture of replaying and enqueueing
    ranks.  Same result either way; describe() and kmc_sampler_launch_mode() report what runs."""
    monkeypatch.delenv("KMC_LAUNCH", raising=False)         # (a forced launch mode would decide instead of the vote)
    nw, nd, G, nburn, seed = 2048, 32, 150, 30, 5
    th = np.random.default_rng(9).standard_normal((nw, nd))
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, nburn, 1, 2.0, seed, nthreads=8), th, store_chain=False)
    for vote in (True, False):
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, nburn, 1, 2.0, seed, moments=True, shard_rank=0, shard_count=1) as s:
            s.rccl_init(kmc.Sampler.rccl_unique_id())
            assert s.rccl_capture() is True                 # RCCL 2.26 accepts the capture of its all-gathers
            s.rccl_set_capture(vote)
            s.set_positions(th)
            s.run(G)
            s.sync()
            how, (mode, _) = s.describe(), s.launch_mode()
            assert ("captured in the graph" in how) == vote and ("launch by launch" in how) == (not vote), how
            assert (mode == 2) == (not vote)
            np.testing.assert_array_equal(s.positions(), ref["final_pos"])
            np.testing.assert_array_equal(s.naccept(), ref["naccept"])
            if not vote:
                with pytest.raises(kmc.KmcError, match="holds no captured chunk"):
                    s.rccl_set_capture(True)                # the chunk is gone: a rank cannot go back alone


def test_device_free_bytes(kmc):
    import ctypes as C
    from kissmcmc_jl_amd import _lib
    free, total = C.c_uint64(0), C.c_uint64(0)
    _lib.check(_lib.lib().kmc_device_free_bytes(0, C.byref(free), C.byref(total)))
    assert 0 < free.value <= total.value and total.value > 200 * 2 ** 30       # 288 GB of HBM3E
    assert _lib.lib().kmc_device_free_bytes(99, C.byref(free), C.byref(total)) == _lib.ERR_BAD_ARG
