"""Walker-sharded emcee across the GPUs of one node: one process per GPU, ``torch.distributed``
(backend ``nccl`` = RCCL over xGMI) for the one real exchange step of the algorithm.

Partition.  With ``h = nwalkers/2`` and ``P`` ranks, rank ``r`` updates the slice
``[r*h/P, (r+1)*h/P)`` of EACH half, so a walker's half -- and therefore its role in the
half-split of reference ``src/samplers.jl:247`` -- is the same as in the 1-GPU indexing.  Every
rank keeps a full ``[nwalkers][ndim]`` replica; during a half-step the complementary half is
read-only (partners, ``src/samplers.jl:250,255``), and after it the updated slices of the active
half are all-gathered in place.  The random stream is keyed by the GLOBAL walker index, so a
P-rank run is bit-identical to the 1-rank run.

The compute of a half-step is delegated to an *executor*:

* :class:`HipShardExecutor` -- the product path: a :class:`~.sampler.Sampler` created with
  ``shard_rank/shard_count`` whose position buffer is a torch CUDA tensor (zero-copy for RCCL).
* tests inject a CPU executor (backed by the oracle) to exercise the partition / exchange logic
  over ``gloo`` without a GPU.
"""
from __future__ import annotations

import numpy as np

try:  # torch is plumbing here: device memory, streams, collectives
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def shard_slice(nwalkers: int, rank: int, world: int):
    """``(begin, count)`` of the slice of EACH half that ``rank`` updates."""
    h = nwalkers // 2
    if nwalkers % 2 != 0:
        raise AssertionError("Use an even number of walkers.")
    if h % world != 0:
        raise ValueError(f"nwalkers/2 = {h} must be divisible by the number of ranks ({world})")
    n = h // world
    return rank * n, n


class HipShardExecutor:
    """Half-steps of this rank's slice on its MI355X, state in a torch tensor."""

    def __init__(self, pdf, nwalkers, ndim, ngenerations, nburnin=0, nthin=1, a_scale=2.0, seed=0,
                 rank=0, world=1, device=None, moments=True):
        from .sampler import Sampler
        if torch is None or not torch.cuda.is_available():
            raise RuntimeError("HipShardExecutor needs a HIP device (torch.cuda); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self.pos = torch.empty((self.nwalkers, self.ndim), dtype=torch.float64, device=self.device)
        self.sampler = Sampler(pdf, nwalkers, ndim, ngenerations, nburnin, nthin, a_scale, seed,
                               moments=moments, use_graph=False, device=self.device.index,
                               shard_rank=rank, shard_count=world)
        self.sampler.bind_positions(self.pos.data_ptr())
        self.sampler.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def set_positions(self, theta):
        self.sampler.set_positions(theta)

    def half_step(self, generation: int, half: int):
        assert generation == self.sampler.generation
        self.sampler.half_step(half)

    def half_view(self, half: int):
        h = self.nwalkers // 2
        return self.pos[half * h:(half + 1) * h]

    def sync(self):
        self.sampler.sync()

    def positions(self):
        return self.pos.cpu().numpy()

    def local_logp(self):
        return torch.from_numpy(self.sampler.logp()).to(self.device)

    def local_naccept(self):
        return torch.from_numpy(self.sampler.naccept()).to(self.device)

    def local_moments(self):
        s, q, n = self.sampler.moments()
        return torch.from_numpy(np.concatenate([s, q, [float(n)]])).to(self.device)

    def close(self):
        self.sampler.close()


def local_to_global(local, nwalkers: int, world: int):
    """Assemble per-rank arrays in P2P local order (first-half slice, then second-half slice) into
    the global walker order.  ``local``: list over ranks of arrays ``[2*h/world, ...]``."""
    h = nwalkers // 2
    n = h // world
    first = [np.asarray(a)[:n] for a in local]
    second = [np.asarray(a)[n:] for a in local]
    return np.concatenate(first + second, axis=0)


def _raise_together(votes, what: str, own_error=None):
    """`votes[r]` = None, or rank r's error text (already gathered from every rank): when any rank failed, raise on THIS rank too."""
    bad = [(r, v) for r, v in enumerate(votes) if v is not None]
    if bad:
        msg = f"{what} failed on rank(s) " + "; ".join(f"{r}: {v}" for r, v in bad)
        raise RuntimeError(msg) from own_error


class P2PEmcee:
    """Walker-sharded emcee with peer-to-peer partner reads (``KMC_P2P``): each rank holds only its
    own walkers, the half-step kernel reads partner rows straight from the owning GPU's HBM over
    xGMI, and half-steps are ordered by per-rank progress flags written by tiny signal kernels --
    no host involvement and no collective on the data path: the whole run is enqueued (hipGraph
    replay) like the single-GPU case.  Compared with all-gathering the updated half after every
    half-step this moves only the rows that are actually drawn (1/P of the bytes).

    Variants of the exchange (both bit-identical, both reading partner rows with system-scope loads; ``bench.py`` checks and times both and runs the faster):
    the default pulls every drawn row from its owner; ``push`` keeps local copies of the other shards and writes accepted rows to every peer
    (``KMC_P2P_PUSH``: fewer bytes per link while the acceptance is below 1 / world size); ``finegrained`` puts the rows in fine-grained memory.

    ``torch.distributed`` (any backend) is used for the rendezvous (IPC handle exchange, barriers)
    and for assembling results.
    """

    def __init__(self, pdf, nwalkers, ndim, ngenerations, nburnin=0, nthin=1, a_scale=2.0, seed=0,
                 device=0, moments=True, group=None, use_graph=True, finegrained=False, push=False,
                 store_chain=False, store_logp=False, connect=True):
        """``connect=False``: only this rank's LOCAL set-up (sampler, buffers) -- a driver that must survive one rank failing
        here votes on the outcome before it calls :meth:`connect`, the collective part (handle exchange + IPC open)."""
        from .sampler import Sampler
        self.group = group
        self.rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        shard_slice(self.nwalkers, self.rank, self.world)
        self.sampler = Sampler(pdf, nwalkers, ndim, ngenerations, nburnin, nthin, a_scale, seed,
                               moments=moments, use_graph=use_graph, device=device, store_chain=store_chain, store_logp=store_logp,
                               shard_rank=self.rank, shard_count=self.world, p2p=True, p2p_finegrained=finegrained,
                               p2p_push=push)
        self._handle = self.sampler.p2p_export() if self.world > 1 else None      # (local: the IPC handles of this rank's buffers)
        if connect:
            self.connect()

    def connect(self):
        """Collective: exchange the IPC handles and open the peers' buffers.  A rank whose part fails (``hipIpcOpenMemHandle``
        refusing a handle, ...) still reaches the vote that follows, and then EVERY rank raises: nobody is left waiting in the
        next collective for a rank that has already given up."""
        if self.world > 1:
            blobs = [None] * self.world
            dist.all_gather_object(blobs, self._handle, group=self.group)
            err = None
            try:
                self.sampler.p2p_connect(blobs)
            except Exception as e:  # noqa: BLE001
                err = e
            votes = [None] * self.world
            dist.all_gather_object(votes, None if err is None else f"{type(err).__name__}: {err}", group=self.group)
            _raise_together(votes, "P2PEmcee.connect (kmc_sampler_p2p_connect)", err)

    def _barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)

    def link_probe(self, nrows=None, reps=20):
        """Collective: every rank measures the link to its next peer (and its own memory) with the pull's access pattern while nobody
        samples: {"link_gather_GBs", "link_copy_GBs", "local_gather_GBs", "rows"} of THIS rank (``Sampler.p2p_link_probe``); a rank whose
        probe fails still reaches every barrier and returns {"error": ...}."""
        nrows = int(nrows or self.nwalkers // self.world // 2 * max(1, self.world - 1))        # the remote rows one half-step draws
        out = {"rows": nrows}
        try:
            self.sampler.sync()
        except Exception as e:  # noqa: BLE001  (nothing may keep this rank from the barriers below)
            out["error"] = f"{type(e).__name__}: {e}"
        self._barrier()                 # nobody is sampling (or writing rows) while the links are measured
        for key, peer in (("link", (self.rank + 1) % self.world), ("local", self.rank)):     # (the local figure without a peer reading this rank's memory)
            try:
                if "error" in out:
                    raise RuntimeError(out["error"])
                g, c = self.sampler.p2p_link_probe(peer, nrows, reps)
                out[key + "_gather_GBs"] = g
                if key == "link":
                    out["link_copy_GBs"] = c
            except Exception as e:  # noqa: BLE001
                out["error"] = f"{type(e).__name__}: {e}"
            self._barrier()
        return out

    def set_positions(self, theta_global):
        self.sampler.sync()             # this rank's own half-steps have drained (an asynchronous run() may precede) ...
        self._barrier()                 # ... and so have everybody else's: nobody is still reading this rank's rows or flags
        self.sampler.set_positions(theta_global)
        self._barrier()                 # every rank's rows are in place and its flags are zero

    def run(self, ngenerations: int):
        self.sampler.run(ngenerations)  # asynchronous; ordering across ranks happens on the devices

    def sync(self):
        self.sampler.sync()

    def close(self):
        self.sync()
        self._barrier()                 # peers may still be reading this rank's rows
        self.sampler.close()

    # -- results ----------------------------------------------------------------------------
    def _gather(self, local):
        if self.world == 1:
            return local_to_global([local], self.nwalkers, 1)
        parts = [None] * self.world
        dist.all_gather_object(parts, local, group=self.group)
        return local_to_global(parts, self.nwalkers, self.world)

    def positions(self):
        return self._gather(self.sampler.positions())

    def logp(self):
        return self._gather(self.sampler.logp())

    def naccept(self):
        return self._gather(self.sampler.naccept())

    def moments(self):
        s, q, n = self.sampler.moments()
        if self.world > 1:
            parts = [None] * self.world
            dist.all_gather_object(parts, (s, q, n), group=self.group)
            s = sum(p[0] for p in parts)
            q = sum(p[1] for p in parts)
            n = sum(p[2] for p in parts)
        return s, q, n

    def local_chain(self, logp=True):
        """This rank's walkers' samples in the reference's order, ``(thetas [nlocal, k, ndim], logdensities [nlocal, k] | None)``
        (local order: its slice of the first half, then of the second half; transposed on the device)."""
        return self.sampler.chain(logp=logp, by_walker=True)

    def gather_chain(self, logp=True):
        """The whole chain by walker, ``(thetas [nwalkers, k, ndim], logdensities [nwalkers, k] | None)`` in global walker
        order, identical on every rank (every rank receives everybody's samples: chains that fit one host, and tests)."""
        ch, lp = self.local_chain(logp)
        return self._gather(ch), (self._gather(lp) if lp is not None else None)


def emcee_p2p(pdf, theta0s, niter: int = 10 ** 5, nburnin=None, nthin: int = 1, a_scale: float = 2.0, seed: int = 0,
              device=None, group=None, gather: bool = True):
    """``emcee`` (reference ``src/samplers.jl:188-293``) with the walkers sharded over the ranks of a ``torch.distributed`` job,
    the reference's partner rule exactly (bit-identical to the one-GPU run with the same ``seed``): call it on every rank with
    the same arguments.  Returns the reference's tuple ``(thetas, accept_ratio, logdensities, None)`` in global walker order,
    identical on every rank (``gather=False``: this rank's walkers only, in its local order)."""
    from .api import emcee_counts
    th = np.asarray(theta0s, dtype=np.float64)
    scalar = th.ndim == 1
    th = th.reshape(th.shape[0], -1)
    nw, nd = th.shape
    if not a_scale > 1:
        raise AssertionError("a_scale>1")
    if nw % 2 != 0:
        raise AssertionError("Use an even number of walkers.")
    G, nburn, ns = emcee_counts(niter, nw, nburnin, nthin)
    if nw < nd + 2:
        raise AssertionError("Use more walkers: at least DOF+2, but better many more.")
    if device is None:
        device = torch.cuda.current_device() if torch is not None and torch.cuda.is_available() else 0
    drv = P2PEmcee(pdf, nw, nd, G, nburn, nthin, a_scale, seed, device=device, moments=False, group=group, store_chain=True, store_logp=True)
    try:
        drv.set_positions(th)
        drv.run(G)
        drv.sync()
        denom = float(G - nburn)
        if gather:
            thetas, logd = drv.gather_chain()
            acc = drv.naccept() / denom if G > nburn else np.full(nw, np.nan)                # :291
        else:
            thetas, logd = drv.local_chain()
            acc = drv.sampler.naccept() / denom if G > nburn else np.full(thetas.shape[0], np.nan)
    finally:
        drv.close()
    if scalar:
        thetas = thetas[:, :, 0]
    return thetas, acc, logd, None


class AllGatherEmcee:
    """Walker-sharded emcee with the exchange the north star names -- an RCCL all-gather of the updated half after every
    half-step (the join of ``src/samplers.jl:273``, across GPUs) -- driven natively: every rank's sampler holds a full
    replica, updates its slice of each half, and ``kmc_sampler_run`` enqueues kernel + in-place ``ncclAllGather`` per
    half-step on one stream (inside the hipGraph chunks where RCCL allows capture).  No host loop, no Python between
    half-steps.  ``torch.distributed`` (any backend) only carries the RCCL unique id and assembles results.
    Bit-identical to the unsharded run (RNG keyed by the global walker index)."""

    def __init__(self, pdf, nwalkers, ndim, ngenerations, nburnin=0, nthin=1, a_scale=2.0, seed=0, device=0, moments=True,
                 group=None, use_graph=True, connect=True):
        """``connect=False``: only this rank's local set-up (its replica sampler); :meth:`connect` is the collective part (unique id
        broadcast, ``ncclCommInitRank``, the vote on the graph capture)."""
        from .sampler import Sampler
        self.group = group
        self.rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self.begin, self.count = shard_slice(self.nwalkers, self.rank, self.world)
        self.sampler = Sampler(pdf, nwalkers, ndim, ngenerations, nburnin, nthin, a_scale, seed, moments=moments,
                               use_graph=use_graph, device=device, shard_rank=self.rank, shard_count=self.world)
        self._use_graph = bool(use_graph)
        self.captured = False
        if connect:
            self.connect()

    def connect(self):
        """Collective over all ranks.  Every step that can fail on ONE rank (loading librccl for the unique id, ``ncclCommInitRank``,
        the graph capture) is followed by a vote every rank reaches, and a failure anywhere raises on EVERY rank -- the library
        call itself never leaves the other ranks blocked in a collective."""
        from .sampler import Sampler
        group, use_graph = self.group, self._use_graph
        uid, err = [None, None], None
        if self.rank == 0:
            try:
                uid[0] = Sampler.rccl_unique_id()
            except Exception as e:  # noqa: BLE001
                err = e
                uid[1] = f"{type(e).__name__}: {e}"
        if self.world > 1:
            dist.broadcast_object_list(uid, src=0, group=group)
        if uid[0] is None:
            raise RuntimeError(f"AllGatherEmcee.connect: rank 0 could not create the RCCL unique id ({uid[1]})") from err
        got = False
        try:
            self.sampler.rccl_init(uid[0])
            # captured all-gathers on every rank, or launch by launch on every rank: each rank captures its chunk now and the
            # outcomes are reduced (MIN) before the first replay -- a rank never decides alone
            got = self.sampler.rccl_capture() if use_graph else False
        except Exception as e:  # noqa: BLE001
            err = e
        if self.world > 1:
            votes = [None] * self.world
            dist.all_gather_object(votes, (None if err is None else f"{type(err).__name__}: {err}", bool(got)), group=group)
            _raise_together([v[0] for v in votes], "AllGatherEmcee.connect (kmc_sampler_rccl_init / _rccl_capture)", err)
            agreed = all(v[1] for v in votes)
        else:
            if err is not None:
                raise err
            agreed = got
        self.captured = bool(agreed)
        if use_graph:
            self.sampler.rccl_set_capture(self.captured)

    def set_positions(self, theta_global):
        self.sampler.set_positions(theta_global)          # every rank: the whole ensemble (its replica)
        if self.world > 1:
            dist.barrier(group=self.group)

    def run(self, ngenerations: int):
        self.sampler.run(ngenerations)

    def sync(self):
        self.sampler.sync()

    def _own(self, a):
        """Keep this rank's slices of a per-walker array, zero the rest; a SUM over ranks assembles it."""
        h = self.nwalkers // 2
        out = np.zeros_like(a)
        for half in (0, 1):
            sl = slice(half * h + self.begin, half * h + self.begin + self.count)
            out[sl] = a[sl]
        return out

    def _sum(self, a):
        if self.world == 1:
            return a
        parts = [None] * self.world
        dist.all_gather_object(parts, a, group=self.group)
        return sum(parts)

    def positions(self):
        return self.sampler.positions()                   # the replica is complete on every rank

    def logp(self):
        return self._sum(self._own(self.sampler.logp()))

    def naccept(self):
        return self._sum(self._own(self.sampler.naccept()))

    def moments(self):
        s, q, n = self.sampler.moments()
        if self.world > 1:
            parts = [None] * self.world
            dist.all_gather_object(parts, (s, q, n), group=self.group)
            s, q, n = sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts)
        return s, q, n

    def close(self):
        self.sync()
        if self.world > 1:
            dist.barrier(group=self.group)
        self.sampler.close()


class ShardedEmcee:
    """The generation loop of ``_emcee`` (``src/samplers.jl:245-290``) over ``world`` ranks, the exchange as a torch
    collective per half-step from Python (any backend: the CPU tests run it over gloo with the oracle as compute
    stand-in; on GPUs :class:`AllGatherEmcee` is the native form of the same exchange)."""

    def __init__(self, executor, nwalkers: int, ndim: int, group=None):
        self.ex = executor
        self.group = group
        self.rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self.begin, self.count = shard_slice(self.nwalkers, self.rank, self.world)
        self.generation = 0

    def _exchange(self, half: int):
        """All-gather the slices of the half that was just updated (in place)."""
        if self.world == 1:
            return
        full = self.ex.half_view(half)                          # [h, ndim], contiguous
        mine = full[self.begin:self.begin + self.count]         # this rank's chunk of it
        dist.all_gather_into_tensor(full, mine, group=self.group)

    def run(self, ngenerations: int):
        for _ in range(int(ngenerations)):
            for half in (0, 1):                                 # :246-247
                self.ex.half_step(self.generation, half)
                self._exchange(half)                            # the join of :273, across ranks
            self.generation += 1

    # -- results (collectives off the data path) ---------------------------------------------
    def positions(self):
        return self.ex.positions()

    def _sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def _own_only(self, t):
        """Zero everything this rank does not own, so a SUM all-reduce assembles the array."""
        h = self.nwalkers // 2
        mask = torch.zeros(self.nwalkers, dtype=torch.bool, device=t.device)
        for half in (0, 1):
            mask[half * h + self.begin: half * h + self.begin + self.count] = True
        return torch.where(mask, t, torch.zeros_like(t))

    def logp(self):
        return self._sum(self._own_only(self.ex.local_logp()))

    def naccept(self):
        return self._sum(self._own_only(self.ex.local_naccept()))

    def moments(self):
        v = self._sum(self.ex.local_moments())
        nd = self.ndim
        return v[:nd], v[nd:2 * nd], int(round(v[2 * nd]))


# ------------------------------------------------------------------------------------------------------------------
# Dealt sub-ensembles: the multi-GPU mode without a per-half-step exchange (opt-in; include/kissmcmc_hip.h)
# ------------------------------------------------------------------------------------------------------------------
class HipDealExecutor:
    """This rank's sub-ensemble on its MI355X: an ordinary :class:`~.sampler.Sampler` (``deal_rank/deal_count``) running on
    torch's current stream, plus the two exchange buffers as torch tensors (what RCCL sends from / receives into)."""

    def __init__(self, pdf, nsub_walkers, ndim, ngenerations, nburnin=0, nthin=1, a_scale=2.0, seed=0, rank=0, world=1,
                 device=None, moments=True, store_chain=False, store_logp=False):
        from .sampler import Sampler
        if torch is None or not torch.cuda.is_available():
            raise RuntimeError("HipDealExecutor needs a HIP device (torch.cuda); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.nwalkers, self.ndim = int(nsub_walkers), int(ndim)
        self.sampler = Sampler(pdf, nsub_walkers, ndim, ngenerations, nburnin, nthin, a_scale, seed, moments=moments,
                               store_chain=store_chain, store_logp=store_logp, device=self.device.index, deal_rank=rank, deal_count=world)
        self.seed, self.nburnin, self.nthin = int(seed), int(nburnin), int(nthin)
        # a stream of its own (graph capture is not allowed on the legacy default stream); DealtEmcee issues the
        # collective inside `torch.cuda.stream(self.stream)`, so RCCL orders itself against the sampler's kernels
        self.stream = torch.cuda.Stream(self.device)
        self.sampler.set_stream(self.stream.cuda_stream)
        self.send = torch.empty((self.nwalkers, self.ndim + 2), dtype=torch.float64, device=self.device)
        self.recv = torch.empty_like(self.send)

    def set_positions(self, theta_sub):
        self.sampler.set_positions(theta_sub)

    def run(self, ngenerations):
        self.sampler.run(ngenerations)

    def pack(self, epoch):
        self.sampler.deal_pack(epoch, self.send.data_ptr())
        return self.send

    def unpack(self, recv):
        self.sampler.deal_unpack(recv.data_ptr())

    def sync(self):
        self.sampler.sync()

    def results(self):
        """``(walker_ids, positions, logp, naccept, (sum, sumsq, n))`` of this sub-ensemble's slots."""
        s = self.sampler
        mom = s.moments() if s.cfg.flags & 4 else (np.zeros(self.ndim), np.zeros(self.ndim), 0)      # (KMC_MOMENTS)
        return s.walker_ids(), s.positions(), s.logp(), s.naccept(), mom

    def state(self):
        return self.sampler.state()

    def restore(self, state, ids):
        self.sampler.restore(state)
        self.sampler.set_walker_ids(ids)

    def chain(self):
        """``(chain [nsamples_done, S, ndim] | None, chain_logp [nsamples_done, S] | None)`` BY SLOT (see DealtEmcee.chain)."""
        store_chain = bool(self.sampler.cfg.flags & 1)
        store_logp = bool(self.sampler.cfg.flags & 2)
        if not (store_chain or store_logp):
            return None, None
        ch, lp = self.sampler.chain(logp=store_logp) if store_chain else (None, None)
        if not store_chain:
            raise RuntimeError("store_logp without store_chain is not supported by the dealt driver")
        return ch, lp

    def close(self):
        self.sampler.close()


def deal_slot_ids(seed: int, world: int, nsub: int, nepochs: int):
    """Which walker every slot holds during epoch e, ``[nepochs + 1, world * nsub]`` (slot = r * nsub + j; row 0: the initial
    deal, walker = slot): the deals are a pure function of ``(seed, epoch, sub-ensemble)`` (``kmc_deal_perm``), so every rank
    can replay all of them on the host -- no device traffic, no collective."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    S, P = int(nsub), int(world)
    c = S // P
    ids = np.empty((int(nepochs) + 1, P * S), dtype=np.int64)
    ids[0] = np.arange(P * S)
    j = np.arange(S, dtype=np.int64)
    for e in range(int(nepochs)):
        for r in range(P):
            a, cc = C.c_int64(), C.c_int64()
            _lib.check(L.kmc_deal_perm(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), e, r, S, C.byref(a), C.byref(cc)))
            t = (a.value * j + cc.value) % S                          # (S < 2^31: no overflow in 64 bits)
            ids[e + 1, (t // c) * S + r * c + t % c] = ids[e, r * S + j]
    return ids


class DealtEmcee:
    """``world`` sub-ensembles (one per rank / GPU), each running the reference's algorithm unchanged on its own
    walkers for ``epoch_gens`` generations (``src/samplers.jl:245-274``; no fabric traffic, hipGraph replay as on one
    GPU), then ONE ``all_to_all_single`` (RCCL over xGMI) re-deals the walkers across the ranks by a state-independent
    permutation.  Same target distribution as ``emcee``; the partner pool (this rank's complementary half instead of
    the whole ensemble's, ``:250``) is what differs -- opt-in, never what ``bench.py`` reports as ``value``.

    Everything is enqueued: the sampler runs on torch's current stream, so kernels, pack, the collective and unpack are
    ordered on the device and the host never waits inside :meth:`run`.  With a backend that cannot move device
    memory (``gloo``: CPU tests, several ranks sharing one GPU) the exchange is staged through the host.
    """

    def __init__(self, executor, nwalkers_total: int, ndim: int, epoch_gens: int, group=None, always_collective: bool = False):
        self.ex = executor
        self.group = group
        self.always_collective = bool(always_collective)   # world size 1: still go through all_to_all_single (tests of the RCCL path)
        self.rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self.nwalkers, self.ndim, self.epoch_gens = int(nwalkers_total), int(ndim), int(epoch_gens)
        if self.nwalkers % self.world != 0 or (self.nwalkers // self.world) % self.world != 0:
            raise ValueError("nwalkers must be divisible by world^2 (equal chunks in the all-to-all)")
        if self.epoch_gens < 1:
            raise ValueError("epoch_gens must be >= 1")
        self.nsub = self.nwalkers // self.world
        self.generation = 0
        self.deals = 0

    def set_positions(self, theta_global):
        """Sub-ensemble r starts with global walkers ``[r S, (r+1) S)``."""
        th = np.asarray(theta_global, dtype=np.float64).reshape(self.nwalkers, self.ndim)
        self.ex.set_positions(th[self.rank * self.nsub:(self.rank + 1) * self.nsub])
        self.generation = 0
        self.deals = 0

    def _deal(self, epoch: int):
        stream = getattr(self.ex, "stream", None)
        if stream is not None:
            with torch.cuda.stream(stream):
                self._deal_on_current_stream(epoch)
        else:
            self._deal_on_current_stream(epoch)

    def _deal_on_current_stream(self, epoch: int):
        send = self.ex.pack(epoch)
        if self.world == 1 and not (self.always_collective and dist is not None and dist.is_initialized()):
            self.ex.unpack(send)                         # the shuffle alone
        else:
            backend = dist.get_backend(self.group)
            if backend == "nccl" or not send.is_cuda:
                recv = self.ex.recv
                dist.all_to_all_single(recv, send, group=self.group)
            else:                                        # e.g. gloo with device tensors: stage through the host
                h_send = send.cpu()
                h_recv = torch.empty_like(h_send)
                dist.all_to_all_single(h_recv, h_send, group=self.group)
                recv = self.ex.recv
                recv.copy_(h_recv)
            self.ex.unpack(recv)
        self.deals += 1

    def run(self, ngenerations: int):
        n = int(ngenerations)
        while n > 0:
            k = min(n, self.epoch_gens - self.generation % self.epoch_gens)
            self.ex.run(k)
            self.generation += k
            n -= k
            if self.generation % self.epoch_gens == 0:   # after generation g with (g + 1) % E == 0
                self._deal(self.generation // self.epoch_gens - 1)

    def sync(self):
        self.ex.sync()

    # -- results, in GLOBAL WALKER order (collectives off the data path) ------------------------------------------
    def results(self):
        """``dict(positions, logp, naccept, sum, sumsq, n)``; per-walker arrays indexed by the walker's index in the
        initial ensemble, identical on every rank."""
        ids, pos, logp, nacc, (s, q, n) = self.ex.results()
        parts = [(ids, pos, logp, nacc, s, q, n)]
        if self.world > 1:
            parts = [None] * self.world
            dist.all_gather_object(parts, (ids, pos, logp, nacc, s, q, n), group=self.group)
        P = np.empty((self.nwalkers, self.ndim))
        L = np.empty(self.nwalkers)
        A = np.empty(self.nwalkers, dtype=np.int64)
        for i, p, l, a, *_ in parts:
            P[i], L[i], A[i] = p, l, a
        return dict(positions=P, logp=L, naccept=A, sum=sum(p[4] for p in parts), sumsq=sum(p[5] for p in parts),
                    n=int(sum(p[6] for p in parts)))

    def state(self):
        """Checkpoint of THIS rank's sub-ensemble (slots' positions, log-pdfs, counters, the generation); every rank saves its
        own.  The slot -> walker map is not stored: it is a function of the generation's epoch."""
        return self.ex.state()

    def restore(self, state):
        """Resume from :meth:`state` (same configuration, seed, world size, epoch length): the continued run equals the
        uninterrupted one.  Moments restart at the restored generation."""
        g = int(state["generation"])
        epoch = g // self.epoch_gens                     # deals done so far
        ids = deal_slot_ids(self.ex.seed, self.world, self.nsub, epoch)[epoch]
        self.ex.restore(state, ids[self.rank * self.nsub:(self.rank + 1) * self.nsub])
        self.generation = g
        self.deals = epoch

    def chain(self):
        """This rank's stored samples: ``(chain [k, S, ndim], chain_logp [k, S] | None, walker [k, S])`` -- the chain is kept
        BY SLOT (what the kernels write, coalesced), ``walker[i, j]`` is the global index of the walker slot ``j`` held when
        sample ``i`` was taken (replayed from the deal permutations, :func:`deal_slot_ids`).  Pooling the samples of all
        ranks needs no identities (``squash_walkers`` pools them anyway, ``src/samplers.jl:395-413``); per-walker series
        (autocorrelation) take them."""
        ch, lp = self.ex.chain()
        if ch is None:
            return None, None, None
        k = ch.shape[0]
        nburnin, nthin = self.ex.nburnin, self.ex.nthin
        gens = nburnin + (np.arange(k, dtype=np.int64) + 1) * nthin - 1          # generation whose end state sample i is
        epochs = gens // self.epoch_gens
        hist = deal_slot_ids(self.ex.seed, self.world, self.nsub, int(epochs.max()) + 1 if k else 0)
        walker = hist[epochs][:, self.rank * self.nsub:(self.rank + 1) * self.nsub]
        return ch, lp, walker

    def gather_chain(self):
        """The whole chain BY WALKER, ``(thetas [nwalkers, k, ndim], logdensities [nwalkers, k])`` as ``emcee`` returns it,
        identical on every rank -- every rank receives everybody's samples (all_gather_object): for ensembles whose chain
        fits one host, and for tests."""
        ch, lp, walker = self.chain()
        parts = [(ch, lp, walker)]
        if self.world > 1:
            parts = [None] * self.world
            dist.all_gather_object(parts, (ch, lp, walker), group=self.group)
        k = ch.shape[0]
        T = np.empty((self.nwalkers, k, self.ndim))
        Lg = np.empty((self.nwalkers, k)) if lp is not None else None
        rows = np.arange(k)[:, None]
        for c_, l_, w_ in parts:
            T[w_, rows] = c_
            if Lg is not None:
                Lg[w_, rows] = l_
        return T, Lg

    def close(self):
        self.sync()
        if self.world > 1:
            dist.barrier(group=self.group)
        self.ex.close()


class LocalDealtEmcee:
    """All ``P`` sub-ensembles of :class:`DealtEmcee` in ONE process (``executors[r]`` = sub-ensemble r), the all-to-all
    done with tensor copies: the same algorithm and results, for single-process tests and one-GPU rehearsal."""

    def __init__(self, executors, nwalkers_total: int, ndim: int, epoch_gens: int):
        self.exs = list(executors)
        self.world = len(self.exs)
        self.nwalkers, self.ndim, self.epoch_gens = int(nwalkers_total), int(ndim), int(epoch_gens)
        if self.nwalkers % self.world != 0 or (self.nwalkers // self.world) % self.world != 0:
            raise ValueError("nwalkers must be divisible by world^2")
        self.nsub = self.nwalkers // self.world
        self.generation = 0

    def set_positions(self, theta_global):
        th = np.asarray(theta_global, dtype=np.float64).reshape(self.nwalkers, self.ndim)
        for r, ex in enumerate(self.exs):
            ex.set_positions(th[r * self.nsub:(r + 1) * self.nsub])
        self.generation = 0

    def run(self, ngenerations: int):
        n = int(ngenerations)
        c = self.nsub // self.world
        while n > 0:
            k = min(n, self.epoch_gens - self.generation % self.epoch_gens)
            for ex in self.exs:
                ex.run(k)
            self.generation += k
            n -= k
            if self.generation % self.epoch_gens == 0:
                sends = [ex.pack(self.generation // self.epoch_gens - 1) for ex in self.exs]
                for ex in self.exs:
                    ex.sync()
                for q, ex in enumerate(self.exs):
                    for r in range(self.world):
                        ex.recv[r * c:(r + 1) * c].copy_(sends[r][q * c:(q + 1) * c])
                if torch is not None and torch.cuda.is_available():
                    torch.cuda.synchronize()
                for ex in self.exs:
                    ex.unpack(ex.recv)

    def sync(self):
        for ex in self.exs:
            ex.sync()

    def results(self):
        P = np.empty((self.nwalkers, self.ndim))
        L = np.empty(self.nwalkers)
        A = np.empty(self.nwalkers, dtype=np.int64)
        S = Q = 0.0
        N = 0
        for ex in self.exs:
            i, p, l, a, (s, q, n) = ex.results()
            P[i], L[i], A[i] = p, l, a
            S, Q, N = S + s, Q + q, N + n
        return dict(positions=P, logp=L, naccept=A, sum=S, sumsq=Q, n=int(N))

    def state(self):
        return [ex.state() for ex in self.exs]

    def restore(self, states):
        g = int(states[0]["generation"])
        epoch = g // self.epoch_gens
        ids = deal_slot_ids(self.exs[0].seed, self.world, self.nsub, epoch)[epoch]
        for r, (ex, st) in enumerate(zip(self.exs, states)):
            ex.restore(st, ids[r * self.nsub:(r + 1) * self.nsub])
        self.generation = g

    def gather_chain(self):
        """``(thetas [nwalkers, k, ndim], logdensities [nwalkers, k] | None)`` by walker (see DealtEmcee.chain / gather_chain)."""
        ex0 = self.exs[0]
        parts = [ex.chain() for ex in self.exs]
        ch0 = parts[0][0]
        if ch0 is None:
            return None, None
        k = ch0.shape[0]
        gens = ex0.nburnin + (np.arange(k, dtype=np.int64) + 1) * ex0.nthin - 1
        epochs = gens // self.epoch_gens
        hist = deal_slot_ids(ex0.seed, self.world, self.nsub, int(epochs.max()) + 1 if k else 0)
        T = np.empty((self.nwalkers, k, self.ndim))
        Lg = np.empty((self.nwalkers, k)) if parts[0][1] is not None else None
        rows = np.arange(k)[:, None]
        for r, (c_, l_) in enumerate(parts):
            w_ = hist[epochs][:, r * self.nsub:(r + 1) * self.nsub]
            T[w_, rows] = c_
            if Lg is not None:
                Lg[w_, rows] = l_
        return T, Lg

    def close(self):
        for ex in self.exs:
            ex.close()


def emcee_dealt(pdf, theta0s, niter: int = 10 ** 5, nburnin=None, nthin: int = 1, a_scale: float = 2.0, seed: int = 0,
                epoch_gens: int = 64, group=None, gather: bool = True):
    """``emcee`` (reference ``src/samplers.jl:188-293``, same ``niter`` / ``nburnin`` / ``nthin`` bookkeeping) over the ranks of a
    ``torch.distributed`` job in the dealt-sub-ensemble mode: call it on every rank (``torchrun``, one rank per GPU) with the
    same arguments.  Returns the reference's tuple ``(thetas [nwalkers, nsamples, ndim], accept_ratio [nwalkers],
    logdensities [nwalkers, nsamples], None)``, identical on every rank, with per-walker series re-filed through the deals
    (``gather=False``: this rank's samples by slot instead, ``(chain [k, S, ndim], accept_ratio, chain_logp [k, S], walker
    [k, S])`` -- no collective on the chain).  ``seed`` must be the same on all ranks.  Opt-in: the partner pool is the rank's
    own complementary half (``:250``), the target distribution is unchanged."""
    from .api import emcee_counts
    th = np.asarray(theta0s, dtype=np.float64)
    scalar = th.ndim == 1
    th = th.reshape(th.shape[0], -1)
    nw, nd = th.shape
    if not a_scale > 1:
        raise AssertionError("a_scale>1")
    if nw % 2 != 0:
        raise AssertionError("Use an even number of walkers.")
    G, nburn, ns = emcee_counts(niter, nw, nburnin, nthin)
    world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
    if nw % (world * world) != 0:
        raise ValueError("nwalkers must be divisible by world^2")
    if nw // world < nd + 2:
        raise AssertionError("Use more walkers: at least DOF+2, but better many more.")     # per sub-ensemble
    ex = HipDealExecutor(pdf, nw // world, nd, G, nburn, nthin, a_scale, seed, rank=rank, world=world, moments=False,
                         store_chain=True, store_logp=True)
    drv = DealtEmcee(ex, nw, nd, epoch_gens, group=group)
    try:
        drv.set_positions(th)
        drv.run(G)
        drv.sync()
        res = drv.results()
        acc = res["naccept"] / float(G - nburn) if G > nburn else np.full(nw, np.nan)         # :291
        if not gather:
            ch, lp, walker = drv.chain()
            return ch, acc, lp, walker
        thetas, logd = drv.gather_chain()
    finally:
        drv.close()
    if scalar:
        thetas = thetas[:, :, 0]
    assert thetas.shape[1] == ns
    return thetas, acc, logd, None
