"""The N>1 path on CPU: 2 processes over gloo drive ShardedEmcee with an executor whose
half-step compute is the oracle (test infrastructure standing in for the HIP kernel), proving
that the partition, the per-half-step exchange and the global RNG addressing make a sharded run
bit-identical to the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NW, ND, G, NBURN, SEED = 128, 8, 25, 5, 77


class OracleShardExecutor:
    """CPU stand-in for HipShardExecutor (tests only)."""

    def __init__(self, oracle, cfg, rank, world):
        self.oracle, self.cfg = oracle, cfg
        self.nwalkers, self.ndim = cfg.nwalkers, cfg.ndim
        h = self.nwalkers // 2
        self.begin, self.count = rank * (h // world), h // world
        self.pos = torch.zeros((self.nwalkers, self.ndim), dtype=torch.float64)
        self.logp_ = np.zeros(self.nwalkers)
        self.nacc = np.zeros(self.nwalkers, dtype=np.int64)
        self.msum = np.zeros(self.ndim); self.msq = np.zeros(self.ndim); self.nmom = 0

    def set_positions(self, theta):
        self.pos.copy_(torch.from_numpy(np.asarray(theta, dtype=np.float64)))
        self.logp_[:] = self.oracle.logpdf_batch(self.cfg.density, list(self.cfg.params), self.pos.numpy())

    def half_step(self, generation, half):
        n = generation + 1 - self.cfg.nburnin
        self.oracle.half_step(self.cfg, self.pos.numpy(), self.logp_, self.nacc, generation, half,
                              self.begin, self.count, count_accept=n > 0)
        if n > 0 and half == 1:    # own slices of both halves are final for this generation
            h = self.nwalkers // 2
            for hf in (0, 1):
                rows = self.pos.numpy()[hf * h + self.begin: hf * h + self.begin + self.count]
                self.msum += rows.sum(axis=0); self.msq += (rows ** 2).sum(axis=0); self.nmom += len(rows)

    def half_view(self, half):
        h = self.nwalkers // 2
        return self.pos[half * h:(half + 1) * h]

    def positions(self):
        return self.pos.numpy().copy()

    def local_logp(self):
        return torch.from_numpy(self.logp_.copy())

    def local_naccept(self):
        return torch.from_numpy(self.nacc.copy())

    def local_moments(self):
        return torch.from_numpy(np.concatenate([self.msum, self.msq, [float(self.nmom)]]))


def _theta0():
    return np.random.default_rng(5).standard_normal((NW, ND))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from kissmcmc_jl_amd.distributed import ShardedEmcee
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED)
        ex = OracleShardExecutor(oracle, cfg, rank, world)
        ex.set_positions(_theta0())
        drv = ShardedEmcee(ex, NW, ND)
        drv.run(G)
        s, q, n = drv.moments()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), pos=drv.positions(), logp=drv.logp(),
                 nacc=drv.naccept(), s=s, q=q, n=n)
    finally:
        dist.destroy_process_group()


def _free_port():
    from portpick import rendezvous_port
    return rendezvous_port()


def test_two_rank_run_is_bit_identical_to_one_rank(oracle, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED)
    ref = oracle.emcee(cfg, _theta0(), store_chain=False)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        np.testing.assert_array_equal(z["pos"], ref["final_pos"])     # every rank ends with the full ensemble
        np.testing.assert_array_equal(z["logp"], ref["final_logp"])
        np.testing.assert_array_equal(z["nacc"], ref["naccept"])
        assert int(z["n"]) == ref["nmoment"]
        np.testing.assert_allclose(z["s"], ref["sum"], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(z["q"], ref["sumsq"], rtol=1e-12, atol=1e-10)


def test_shard_slice(kmc):
    from kissmcmc_jl_amd.distributed import shard_slice
    assert shard_slice(65536 * 8, 3, 8) == (3 * 32768, 32768)
    assert shard_slice(100, 0, 1) == (0, 50)
    with pytest.raises(ValueError):
        shard_slice(100, 0, 4)
    with pytest.raises(AssertionError):
        shard_slice(101, 0, 1)


class _FailingSampler:
    """Stands in for the HIP sampler inside P2PEmcee / AllGatherEmcee (no device here): the collective wiring is what is under test."""

    def __init__(self, fail):
        self.fail = fail

    def p2p_connect(self, blobs):
        assert all(b is not None for b in blobs)
        if self.fail:
            raise RuntimeError("hipIpcOpenMemHandle: invalid argument")

    def rccl_init(self, uid):
        assert uid == b"u" * 128
        if self.fail:
            raise RuntimeError("hipStreamBeginCapture: operation not permitted")

    def rccl_capture(self):
        return True

    def rccl_set_capture(self, flag):
        self.captured = flag


def _connect_worker(rank, world, port, outdir, which, failing_rank):
    sys.path.insert(0, ROOT)
    from kissmcmc_jl_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        if which == "p2p":
            d = D.P2PEmcee.__new__(D.P2PEmcee)
            d.group, d.rank, d.world, d._handle = None, rank, world, b"h%d" % rank
        else:
            from kissmcmc_jl_amd import sampler as S
            S.Sampler.rccl_unique_id = staticmethod(lambda: b"u" * 128)
            d = D.AllGatherEmcee.__new__(D.AllGatherEmcee)
            d.group, d.rank, d.world, d._use_graph, d.captured = None, rank, world, True, False
        d.sampler = _FailingSampler(rank == failing_rank)
        try:
            d.connect()
            outcome = "connected"
        except RuntimeError as e:
            outcome = str(e)
        dist.barrier()                       # every rank came out of connect(): nobody is stuck in a collective
        open(os.path.join(outdir, f"{which}{rank}.txt"), "w").write(outcome)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["p2p", "allgather"])
@pytest.mark.parametrize("failing_rank", [-1, 1])
def test_connect_fails_on_every_rank_when_it_fails_on_one(tmp_path, which, failing_rank):
    """P2PEmcee.connect / AllGatherEmcee.connect: a rank whose local part raises (IPC open, ncclCommInitRank, graph capture) must not
    leave the others blocked in the next collective -- every rank learns of it in a vote and raises the same error."""
    mp.spawn(_connect_worker, args=(2, _free_port(), str(tmp_path), which, failing_rank), nprocs=2, join=True)
    got = [open(os.path.join(str(tmp_path), f"{which}{r}.txt")).read() for r in range(2)]
    if failing_rank < 0:
        assert got == ["connected", "connected"]
    else:
        for g in got:
            assert "failed on rank(s) 1:" in g and ("hipIpcOpenMemHandle" in g or "hipStreamBeginCapture" in g)
