"""CPU: the dealt-sub-ensemble mode (the multi-GPU mode without a per-half-step exchange; include/kissmcmc_hip.h,
oracle kmco_emcee_dealt).  The oracle's restatement against its own invariants; the deal permutation of the product
library against the oracle's; and the N>1 driver (distributed.DealtEmcee: epochs + all_to_all_single) over 2 gloo ranks
with the oracle as compute stand-in, bit-identical to the one-process oracle run."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NW, ND, G, NBURN, SEED, E = 256, 6, 40, 10, 99, 7        # G % E != 0: the last epoch is cut short; 5 deals


def _theta0():
    return np.random.default_rng(8).standard_normal((NW, ND))


def test_deal_perm_and_seed_product_equals_oracle(oracle):
    from kissmcmc_jl_amd import _lib
    L = _lib.lib()
    for seed, epoch, rank, S in [(1, 0, 0, 64), (12345, 3, 5, 65536), (2 ** 63 + 5, 2 ** 33, 7, 8192), (7, 1, 2, 96), (9, 4, 0, 2)]:
        a, c = C.c_int64(), C.c_int64()
        assert L.kmc_deal_perm(seed, epoch, rank, S, C.byref(a), C.byref(c)) == 0
        assert (a.value, c.value) == oracle.deal_perm(seed, epoch, rank, S)
        assert np.gcd(a.value, S) == 1 and 0 <= c.value < S                        # a bijection of the S slots
        assert len({(a.value * j + c.value) % S for j in range(min(S, 4096))}) == min(S, 4096)
        assert int(L.kmc_deal_seed(seed, rank)) == oracle.deal_seed(seed, rank) == (seed + (rank + 1) * 0x9E3779B97F4A7C15) % 2 ** 64


def test_oracle_dealt_invariants(oracle):
    th = _theta0()
    cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED)
    r = oracle.emcee_dealt(cfg, 4, E, th)
    assert r["status"] == 0 and sorted(r["slot_ids"]) == list(range(NW)) and r["nmoment"] == NW * (G - NBURN)
    assert not np.array_equal(r["slot_ids"], np.arange(NW))                        # walkers did change sub-ensembles
    assert len({int(i) // (NW // 4) for i in r["slot_ids"][: NW // 4]}) == 4       # ... sub-ensemble 0 ends with walkers of all four
    # one sub-ensemble that is never dealt = the reference's algorithm with that sub-ensemble's key
    r1 = oracle.emcee_dealt(cfg, 1, 10 ** 9, th)
    cfg1 = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, oracle.deal_seed(SEED, 0))
    r2 = oracle.emcee(cfg1, th, store_chain=False)
    np.testing.assert_array_equal(r1["final_pos"], r2["final_pos"])
    np.testing.assert_array_equal(r1["naccept"], r2["naccept"])
    # ... its chain too; and a walker's last stored sample is where it ended (G - NBURN stored, nthin = 1)
    r1c = oracle.emcee_dealt(cfg, 1, 10 ** 9, th, store_chain=True)
    r2c = oracle.emcee(cfg1, th)
    np.testing.assert_array_equal(r1c["chain"], r2c["chain"])
    rc = oracle.emcee_dealt(cfg, 4, E, th, store_chain=True)
    np.testing.assert_array_equal(rc["chain"][-1], rc["final_pos"])
    np.testing.assert_array_equal(rc["chain_logp"][-1], rc["final_logp"])
    np.testing.assert_array_equal(rc["final_pos"], r["final_pos"])
    # every sub-ensemble must itself be a valid emcee ensemble (src/samplers.jl:202-205)
    bad = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], 24, ND, 5, 0, 1, 2.0, SEED)
    assert oracle.emcee_dealt(bad, 4, 2, np.zeros((24, ND)))["status"] == oracle.ERR_TOO_FEW_WALKERS   # S = 6 < ndim + 2


def test_oracle_dealt_samples_the_target(oracle):
    """The deal ignores the state and every sub-ensemble update is an emcee move: same stationary distribution."""
    nw, nd, g = 512, 4, 3000
    th = np.random.default_rng(1).standard_normal((nw, nd))
    cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [-5.0, 3.0], nw, nd, g, g // 3, 1, 2.0, 5, nthreads=4)
    r = oracle.emcee_dealt(cfg, 8, 16, th - 5.0)
    mean = r["sum"] / r["nmoment"]
    var = r["sumsq"] / r["nmoment"] - mean ** 2
    assert np.abs(mean + 5.0).max() < 0.05 * 3 and np.abs(np.sqrt(var) - 3.0).max() < 0.05 * 3
    assert 0.3 < r["accept_ratio"].mean() < 0.8


class OracleDealExecutor:
    """CPU stand-in for HipDealExecutor (tests only): one sub-ensemble, half-steps by the oracle."""

    def __init__(self, oracle, cfg_total, rank, world):
        self.oracle = oracle
        self.S = cfg_total.nwalkers // world
        self.rank, self.world, self.seed0 = rank, world, cfg_total.seed
        self.cfg = oracle.make_config(cfg_total.density, list(cfg_total.params), self.S, cfg_total.ndim, cfg_total.ngenerations,
                                      cfg_total.nburnin, cfg_total.nthin, cfg_total.a_scale, oracle.deal_seed(cfg_total.seed, rank))
        nd = cfg_total.ndim
        self.pos = np.zeros((self.S, nd)); self.logp = np.zeros(self.S); self.nacc = np.zeros(self.S, dtype=np.int64)
        self.ids = np.arange(rank * self.S, (rank + 1) * self.S, dtype=np.int64)
        self.msum = np.zeros(nd); self.msq = np.zeros(nd); self.nmom = 0
        self.seed, self.nburnin, self.nthin = cfg_total.seed, cfg_total.nburnin, cfg_total.nthin      # (what DealtEmcee.chain reads)
        self.chain_rows, self.chain_lp = [], []           # by slot, as the device stores it
        self.gen = 0
        self.recv = torch.zeros((self.S, nd + 2), dtype=torch.float64)

    def set_positions(self, th):
        self.pos[:] = th
        self.logp[:] = self.oracle.logpdf_batch(self.cfg.density, list(self.cfg.params), self.pos)

    def run(self, n):
        for _ in range(n):
            k = self.gen + 1 - self.cfg.nburnin
            for half in (0, 1):
                self.oracle.half_step(self.cfg, self.pos, self.logp, self.nacc, self.gen, half, 0, self.S // 2, count_accept=k > 0)
            if k > 0 and k % self.cfg.nthin == 0:
                self.msum += self.pos.sum(axis=0); self.msq += (self.pos ** 2).sum(axis=0); self.nmom += self.S
                self.chain_rows.append(self.pos.copy()); self.chain_lp.append(self.logp.copy())
            self.gen += 1

    def pack(self, epoch):
        a, c = self.oracle.deal_perm(self.seed0, epoch, self.rank, self.S)
        t = (a * np.arange(self.S) + c) % self.S
        buf = np.zeros((self.S, self.pos.shape[1] + 2))
        buf[t, :-2] = self.pos
        buf[t, -2] = self.logp
        buf[t, -1] = ((self.ids.astype(np.uint64) << np.uint64(32)) | self.nacc.astype(np.uint64)).view(np.float64)
        return torch.from_numpy(buf)

    def unpack(self, recv):
        r = recv.numpy()
        self.pos[:] = r[:, :-2]
        self.logp[:] = r[:, -2]
        w = np.ascontiguousarray(r[:, -1]).view(np.uint64)
        self.nacc[:] = (w & np.uint64(0xFFFFFFFF)).astype(np.int64)
        self.ids[:] = (w >> np.uint64(32)).astype(np.int64)

    def sync(self):
        pass

    def results(self):
        return self.ids.copy(), self.pos.copy(), self.logp.copy(), self.nacc.copy(), (self.msum, self.msq, self.nmom)

    def chain(self):
        return np.array(self.chain_rows).reshape(-1, self.S, self.pos.shape[1]), np.array(self.chain_lp).reshape(-1, self.S)

    def close(self):
        pass


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import oracle
    from kissmcmc_jl_amd.distributed import DealtEmcee
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED)
        drv = DealtEmcee(OracleDealExecutor(oracle, cfg, rank, world), NW, ND, E)
        drv.set_positions(_theta0())
        drv.run(13)                       # in pieces: a run may stop anywhere inside an epoch
        drv.run(G - 13)
        res = drv.results()
        assert drv.deals == G // E
        res["thetas"], res["logd"] = drv.gather_chain()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), **res)
        drv.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    from portpick import rendezvous_port
    return rendezvous_port()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_ranks_equal_the_oracle_run(oracle, tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED)
    ref = oracle.emcee_dealt(cfg, world, E, _theta0(), store_chain=True)
    assert ref["status"] == 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        np.testing.assert_array_equal(z["thetas"], ref["chain"].transpose(1, 0, 2))      # by walker, through every deal
        np.testing.assert_array_equal(z["logd"], ref["chain_logp"].T)
        np.testing.assert_array_equal(z["positions"], ref["final_pos"])
        np.testing.assert_array_equal(z["logp"], ref["final_logp"])
        np.testing.assert_array_equal(z["naccept"], ref["naccept"])
        assert int(z["n"]) == ref["nmoment"]
        np.testing.assert_allclose(z["sum"], ref["sum"], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(z["sumsq"], ref["sumsq"], rtol=1e-12, atol=1e-10)


def test_local_driver_equals_the_oracle_run(oracle):
    """LocalDealtEmcee (all sub-ensembles in one process, copies instead of the collective) is the same algorithm."""
    from kissmcmc_jl_amd.distributed import LocalDealtEmcee
    cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED)
    exs = [OracleDealExecutor(oracle, cfg, r, 4) for r in range(4)]
    drv = LocalDealtEmcee(exs, NW, ND, E)
    drv.set_positions(_theta0())
    drv.run(G)
    res = drv.results()
    ref = oracle.emcee_dealt(cfg, 4, E, _theta0(), store_chain=True)
    np.testing.assert_array_equal(res["positions"], ref["final_pos"])
    np.testing.assert_array_equal(res["naccept"], ref["naccept"])
    assert res["n"] == ref["nmoment"]
    thetas, logd = drv.gather_chain()
    np.testing.assert_array_equal(thetas, ref["chain"].transpose(1, 0, 2))
    np.testing.assert_array_equal(logd, ref["chain_logp"].T)
