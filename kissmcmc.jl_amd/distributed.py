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


class P2PEmcee:
    """Walker-sharded emcee with peer-to-peer partner reads (``KMC_P2P``): each rank holds only its
    own walkers, the half-step kernel reads partner rows straight from the owning GPU's HBM over
    xGMI, and half-steps are ordered by per-rank progress flags written by tiny signal kernels --
    no host involvement and no collective on the data path: the whole run is enqueued (hipGraph
    replay) like the single-GPU case.  Compared with all-gathering the updated half after every
    half-step this moves only the rows that are actually drawn (1/P of the bytes).

    Variants of the exchange (all bit-identical; ``bench.py`` picks by measurement): ``push`` keeps local copies of
    the other shards and writes accepted rows to every peer (``KMC_P2P_PUSH``); ``lazy`` keeps the same copies but
    fills them on demand -- ranks publish their accept bytes, a reader pulls a row only when its copy is older than
    the row's last accept (``KMC_P2P_LAZY``, ``Sampler.p2p_stats()``); ``fold_signal`` lets the half-step kernel
    publish its own progress flag (``KMC_P2P_FOLD_SIGNAL``); ``finegrained`` puts the rows in fine-grained memory.

    ``torch.distributed`` (any backend) is used for the rendezvous (IPC handle exchange, barriers)
    and for assembling results.
    """

    def __init__(self, pdf, nwalkers, ndim, ngenerations, nburnin=0, nthin=1, a_scale=2.0, seed=0,
                 device=0, moments=True, group=None, use_graph=True, finegrained=False, fold_signal=False, push=False, lazy=False):
        from .sampler import Sampler
        self.group = group
        self.rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        shard_slice(self.nwalkers, self.rank, self.world)
        self.sampler = Sampler(pdf, nwalkers, ndim, ngenerations, nburnin, nthin, a_scale, seed,
                               moments=moments, use_graph=use_graph, device=device,
                               shard_rank=self.rank, shard_count=self.world, p2p=True, p2p_finegrained=finegrained,
                               p2p_fold=fold_signal, p2p_push=push, p2p_lazy=lazy)
        if self.world > 1:
            blobs = [None] * self.world
            dist.all_gather_object(blobs, self.sampler.p2p_export(), group=group)
            self.sampler.p2p_connect(blobs)

    def _barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)

    def set_positions(self, theta_global):
        self._barrier()                 # nobody is still reading this rank's rows or flags
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


class ShardedEmcee:
    """The generation loop of ``_emcee`` (``src/samplers.jl:245-290``) over ``world`` ranks."""

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
