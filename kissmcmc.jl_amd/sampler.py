"""Thin object wrapper over the stateful C ABI (``kmc_sampler_*`` in ``include/kissmcmc_hip.h``).

The ensemble lives in HBM for the lifetime of the object; ``run`` enqueues generations of the
reference's ``_emcee`` loop (``src/samplers.jl:245-290``) as HIP kernel launches.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .densities import DeviceLogPdf


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Sampler:
    def __init__(self, pdf: DeviceLogPdf, nwalkers: int, ndim: int, ngenerations: int, nburnin: int = 0,
                 nthin: int = 1, a_scale: float = 2.0, seed: int = 0, store_chain: bool = False,
                 store_logp: bool = False, moments: bool = False, use_graph: bool = True,
                 device: int = 0, shard_rank: int = 0, shard_count: int = 1, p2p: bool = False,
                 island_gens: int = 0, island_size: int = 0, p2p_finegrained: bool = False, p2p_push: bool = False,
                 dtype: str = "f64", deal_rank: int = 0, deal_count: int = 0,
                 stream_chain: bool = False, chain_by_walker: bool = False, store_blobs: bool = False):
        if not isinstance(pdf, DeviceLogPdf):
            raise TypeError(
                "pdf must be a menu log-density (GaussianIso, Exponential, Rosenbrock, LogNormal, MvNormal2), "
                "an ExprDensity (compiled for the device) or a HostLogPdf (any callable, evaluated on the host "
                f"per half-step); got {type(pdf).__name__}.")
        pdf.check_ndim(int(ndim))
        self.pdf = pdf
        self._h = None
        cfg = _lib.Config()
        if dtype not in ("f64", "f32"):
            raise ValueError("dtype must be 'f64' (the reference's Float64) or 'f32' (float rows on the device)")
        cfg.dtype = _lib.F32 if dtype == "f32" else _lib.F64
        cfg.density = pdf.density_id
        p = list(pdf.params()) + [0.0] * 8
        for i in range(8):
            cfg.params[i] = float(p[i])
        cfg.nwalkers, cfg.ndim = int(nwalkers), int(ndim)
        cfg.ngenerations, cfg.nburnin, cfg.nthin = int(ngenerations), int(nburnin), int(nthin)
        cfg.a_scale = float(a_scale)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        flags = 0
        if store_chain:
            flags |= _lib.STORE_CHAIN
        if store_logp:
            flags |= _lib.STORE_LOGP
        if moments:
            flags |= _lib.MOMENTS
        if not use_graph:
            flags |= _lib.NO_GRAPH
        if store_blobs:
            flags |= _lib.STORE_BLOBS      # a CDensity(..., nblob=m): keep the blob of every stored sample (src/samplers.jl:270)
        self.nblob = int(getattr(pdf, "nblob", 0) or 0)
        if stream_chain and (store_chain or store_logp):
            # KMC_STREAM_CHAIN: the chain goes to host arrays block by block while sampling (bounded by host RAM, not HBM)
            flags |= _lib.STREAM_CHAIN
            if chain_by_walker:
                # ... in the reference's order, thetas[w][k]: host arrays [nlocal, nsamples, ndim] / [nlocal, nsamples]
                flags |= _lib.CHAIN_BY_WALKER
        if p2p:
            flags |= _lib.P2P
            if p2p_finegrained:
                flags |= _lib.P2P_FINEGRAINED
            if p2p_push:
                flags |= _lib.P2P_PUSH
        if island_gens:
            # ISLAND MODE: 256-walker islands resident in LDS for `island_gens` generations per launch
            flags |= _lib.ISLANDS
            cfg.island_gens = int(island_gens)
            cfg.island_size = int(island_size)
        cfg.flags = flags
        cfg.device = int(device)
        cfg.shard_rank, cfg.shard_count = int(shard_rank), int(shard_count)
        cfg.deal_rank, cfg.deal_count = int(deal_rank), int(deal_count)   # dealt sub-ensembles (distributed.DealtEmcee)
        cfg.user_density = pdf.user_handle     # runtime-compiled density (ExprDensity) or None
        cb = getattr(pdf, "c_callback", None)  # host-evaluated density (HostLogPdf) or None
        if cb is not None:
            cfg.host_logpdf = C.cast(cb, C.c_void_p)
            if getattr(pdf, "c_accepted", None) is not None:   # accept outcomes back to the host (blobs)
                cfg.host_accepted = C.cast(pdf.c_accepted, C.c_void_p)
        self.cfg = cfg
        self._L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._L.kmc_sampler_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self.nlocal = self.nwalkers // max(1, int(shard_count))
        self.p2p = bool(p2p)
        # rows this object holds: the whole ensemble, or (P2P) this shard's slices of both halves
        self.nrows = self.nlocal if self.p2p else self.nwalkers
        self._host_chain = self._host_logp = None
        self._host_by_walker = False
        if flags & _lib.STREAM_CHAIN and self.nsamples > 0:
            # the destination of the streamed chain: host arrays owned by this object (page-locked in place by the library)
            self._host_by_walker = bool(flags & _lib.CHAIN_BY_WALKER)
            if self._host_by_walker:
                self._host_chain = np.empty((self.nlocal, self.nsamples, self.ndim)) if store_chain else None
                self._host_logp = np.empty((self.nlocal, self.nsamples)) if store_logp else None
            else:
                self._host_chain = np.empty((self.nsamples, self.nlocal, self.ndim)) if store_chain else None
                self._host_logp = np.empty((self.nsamples, self.nlocal)) if store_logp else None
            _lib.check(self._L.kmc_sampler_set_chain_host(self._h, _dp(self._host_chain) if store_chain else None,
                                                          _dp(self._host_logp) if store_logp else None))

    # -- lifecycle --------------------------------------------------------------------------
    def close(self):
        if self._h is not None:
            self._L.kmc_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- control ----------------------------------------------------------------------------
    def set_stream(self, hip_stream: int):
        _lib.check(self._L.kmc_sampler_set_stream(self._h, C.c_void_p(hip_stream)))

    def bind_positions(self, device_ptr: int):
        """Use a caller-owned device buffer (``double [nwalkers][ndim]``) for the ensemble."""
        _lib.check(self._L.kmc_sampler_bind_positions(self._h, C.c_void_p(device_ptr)))

    @staticmethod
    def p2p_connect_local(shards):
        """Wire KMC_P2P samplers that live in this process (``shards[r]`` = shard r, same device) to each other."""
        arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
        for s in shards:
            _lib.check(s._L.kmc_sampler_p2p_connect_local(s._h, arr))

    def p2p_link_probe(self, peer: int, nrows: int, reps: int = 20):
        """(GB/s of whole rows of shard ``peer`` gathered at random row indices with the pull's system-scope loads, GB/s of the runtime's
        copy of that shard): one fabric link -- or local memory for ``peer`` = own rank -- measured; the peers must be idle."""
        g, c = C.c_double(0.0), C.c_double(0.0)
        _lib.check(self._L.kmc_sampler_p2p_link_probe(self._h, int(peer), int(nrows), int(reps), C.byref(g), C.byref(c)))
        return g.value, c.value

    @staticmethod
    def rccl_unique_id() -> bytes:
        """A fresh RCCL unique id (rank 0 creates it; every rank of the communicator gets the same bytes)."""
        buf = C.create_string_buffer(_lib.RCCL_ID_BYTES)
        _lib.check(_lib.lib().kmc_rccl_unique_id(buf))
        return buf.raw

    def rccl_init(self, unique_id: bytes):
        """Replica sharding (``shard_rank / shard_count``, no P2P): attach an RCCL communicator; :meth:`run` then enqueues an
        in-place all-gather of the updated half after every half-step kernel.  Collective over all ``shard_count`` ranks."""
        assert len(unique_id) == _lib.RCCL_ID_BYTES
        buf = C.create_string_buffer(bytes(unique_id), _lib.RCCL_ID_BYTES)
        _lib.check(self._L.kmc_sampler_rccl_init(self._h, buf))

    def rccl_capture(self) -> bool:
        """Capture the chunk of kernels + all-gathers now; whether THIS rank got a graph (reduce over the ranks with MIN,
        then :meth:`rccl_set_capture` on every rank: all replay captured all-gathers, or all enqueue them one by one)."""
        got = C.c_int(0)
        _lib.check(self._L.kmc_sampler_rccl_capture(self._h, C.byref(got)))
        return bool(got.value)

    def rccl_set_capture(self, use_captured: bool):
        _lib.check(self._L.kmc_sampler_rccl_set_capture(self._h, 1 if use_captured else 0))

    @staticmethod
    def rccl_version():
        """``"major.minor.patch"`` of the librccl.so the library resolves (``None`` when it cannot be loaded); no device needed."""
        v = C.c_int(0)
        if _lib.lib().kmc_rccl_version(C.byref(v), None, 0) != _lib.OK:
            return None
        return f"{v.value // 10000}.{v.value // 100 % 100}.{v.value % 100}"

    LAUNCH_MODES = {0: "undecided", 1: "table graph", 2: "eager", 3: "updated graph", 4: "single launch per many generations"}

    def launch_mode(self):
        """``(mode, budget_fallback)``: how :meth:`run` issues the launches (a key of ``LAUNCH_MODES``) and whether the
        sampler is outside the updated-graph mode because the process's budget of graph parameter updates was spent."""
        fb = C.c_int(0)
        return int(self._L.kmc_sampler_launch_mode(self._h, C.byref(fb))), bool(fb.value)

    def p2p_export(self) -> bytes:
        """IPC handle blob of this shard (to be all-gathered across the ranks)."""
        buf = C.create_string_buffer(_lib.P2P_HANDLE_BYTES)
        _lib.check(self._L.kmc_sampler_p2p_export(self._h, buf))
        return buf.raw

    def p2p_connect(self, blobs):
        """``blobs``: the export blobs of all ``shard_count`` ranks, in rank order."""
        raw = b"".join(bytes(b) for b in blobs)
        assert len(raw) == _lib.P2P_HANDLE_BYTES * self.cfg.shard_count
        buf = C.create_string_buffer(raw, len(raw))
        _lib.check(self._L.kmc_sampler_p2p_connect(self._h, buf))

    # -- dealt sub-ensembles (kmc_config.deal_count; see include/kissmcmc_hip.h) ---------------
    def deal_pack(self, epoch: int, send_ptr: int):
        """Enqueue the packing of this sub-ensemble's rows ([nwalkers][ndim + 2] doubles, shuffled for the deal of ``epoch``)."""
        _lib.check(self._L.kmc_sampler_deal_pack(self._h, int(epoch), C.c_void_p(send_ptr)))

    def deal_unpack(self, recv_ptr: int):
        _lib.check(self._L.kmc_sampler_deal_unpack(self._h, C.c_void_p(recv_ptr)))

    def walker_ids(self) -> np.ndarray:
        """Global walker index held by each local slot (row order of positions / naccept)."""
        out = np.empty(self.nrows, dtype=np.int64)
        _lib.check(self._L.kmc_sampler_get_walker_ids(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def set_walker_ids(self, ids):
        """Dealt sub-ensembles, after :meth:`restore`: the global walker each slot holds (``distributed.deal_slot_ids``)."""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64).reshape(self.nrows))
        _lib.check(self._L.kmc_sampler_set_walker_ids(self._h, ids.ctypes.data_as(C.POINTER(C.c_int64))))

    def set_positions(self, theta):
        theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).reshape(self.nwalkers, self.ndim))
        self._check_host(self._L.kmc_sampler_set_positions(self._h, _dp(theta)))

    def init_ball(self, theta0, ball_radius, seed: int = 0, ball_radius_halfing_steps: int = 7, ntries: int = 100):
        """Device-side ``make_theta0s`` (reference ``src/samplers.jl:311-349``): seeded Gaussian ball
        around ``theta0`` with ``pdf > -Inf``, generated and checked on the GPU; the sampler is then
        ready to run (no host ensemble, no H2D copy).

        Differs from the host-side :func:`~kissmcmc_jl_amd.make_theta0s` in two documented ways: the random stream
        (Philox keyed by ``(seed, walker)``, not a numpy generator consumed walker by walker) and the ball's shrink
        factor, which restarts at 1 for every walker here -- the host function follows the reference, where
        ``ball_radius`` is never reset (``:326``) and one walker's retries shrink the ball of all later ones.  The two
        agree in distribution whenever no walker needs more than ``ntries`` draws."""
        th = np.ascontiguousarray(np.broadcast_to(np.asarray(theta0, dtype=np.float64), (self.ndim,)))
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(ball_radius, dtype=np.float64), (self.ndim,)))
        try:
            _lib.check(self._L.kmc_sampler_init_ball(self._h, _dp(th), _dp(r), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                     int(ball_radius_halfing_steps), int(ntries)))
        except _lib.KmcError as e:
            if e.status == _lib.ERR_NONFINITE_LOGP:
                raise RuntimeError(str(e)) from e
            raise

    def state(self):
        """Checkpoint: ``dict(positions, logp, naccept, generation)`` (synchronises)."""
        return dict(positions=self.positions(), logp=self.logp(), naccept=self.naccept(), generation=self.generation)

    def restore(self, state):
        """Resume from :meth:`state` of a sampler with the same configuration and seed: the continued
        run is bit-identical to an uninterrupted one (moments restart at the restored generation)."""
        pos = np.ascontiguousarray(np.asarray(state["positions"], dtype=np.float64).reshape(self.nrows, self.ndim))
        lp = np.ascontiguousarray(np.asarray(state["logp"], dtype=np.float64))
        na = np.ascontiguousarray(np.asarray(state["naccept"], dtype=np.int64))
        _lib.check(self._L.kmc_sampler_set_state(self._h, _dp(pos), _dp(lp), na.ctypes.data_as(C.POINTER(C.c_int64)),
                                                 int(state["generation"])))

    def _check_host(self, status):
        """Re-raise an exception the host log-pdf raised inside the C callback, else check the status."""
        err = getattr(self.pdf, "error", None)
        if status != _lib.OK and err is not None:
            self.pdf.error = None
            raise err
        _lib.check(status)

    def run(self, ngenerations: int):
        self._check_host(self._L.kmc_sampler_run(self._h, int(ngenerations)))

    def half_step(self, half: int):
        _lib.check(self._L.kmc_sampler_half_step(self._h, int(half)))

    def sync(self):
        _lib.check(self._L.kmc_sampler_sync(self._h))

    def last_run_ms(self) -> float:
        ms = C.c_double()
        _lib.check(self._L.kmc_sampler_last_run_ms(self._h, C.byref(ms)))
        return ms.value

    @property
    def generation(self) -> int:
        return int(self._L.kmc_sampler_generation(self._h))

    @property
    def nsamples(self) -> int:
        return int(self._L.kmc_sampler_nsamples(self._h))

    @property
    def launch_count(self) -> int:
        return int(self._L.kmc_sampler_launch_count(self._h))

    def describe(self) -> str:
        """How this sampler executes: kernel family, geometry, exchange scheme."""
        buf = C.create_string_buffer(2048)
        _lib.check(self._L.kmc_sampler_describe(self._h, buf, 2048))
        return buf.value.decode()

    def __repr__(self):
        return f"<Sampler {self.nwalkers}x{self.ndim} {self.pdf!r}: {self.describe()}>"

    def int_acorr(self, c: float = 5.0):
        """Integrated autocorrelation time per dimension of the chain stored so far, computed where it lies (on the
        device): ``(tau[ndim], nsamples / tau)``; see :func:`kissmcmc_jl_amd.int_acorr`."""
        tau = np.zeros(self.ndim)
        conv = np.zeros(self.ndim)
        dp = C.POINTER(C.c_double)
        _lib.check(self._L.kmc_sampler_int_acorr(self._h, float(c), tau.ctypes.data_as(dp), conv.ctypes.data_as(dp)))
        return tau, conv

    def device_ptr(self, which: int) -> int:
        return int(self._L.kmc_sampler_device_ptr(self._h, int(which)) or 0)

    # -- downloads --------------------------------------------------------------------------
    def positions(self) -> np.ndarray:
        out = np.empty((self.nrows, self.ndim))
        _lib.check(self._L.kmc_sampler_get_positions(self._h, _dp(out)))
        return out

    def logp(self) -> np.ndarray:
        out = np.empty(self.nrows)
        _lib.check(self._L.kmc_sampler_get_logp(self._h, _dp(out)))
        return out

    def naccept(self) -> np.ndarray:
        out = np.empty(self.nrows, dtype=np.int64)
        _lib.check(self._L.kmc_sampler_get_naccept(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def accept_ratio(self) -> np.ndarray:
        out = np.empty(self.nrows)
        _lib.check(self._L.kmc_sampler_get_accept_ratio(self._h, _dp(out)))
        return out

    def moments(self):
        """``(sum[ndim], sumsq[ndim], n)`` over the samples that would be stored."""
        s = np.empty(self.ndim)
        q = np.empty(self.ndim)
        n = C.c_int64()
        _lib.check(self._L.kmc_sampler_get_moments(self._h, _dp(s), _dp(q), C.byref(n)))
        return s, q, n.value

    def blobs(self, by_walker: bool = True):
        """Stored blobs of a ``CDensity(..., nblob=m)`` sampler created with ``store_blobs=True``: ``[nwalkers, k, m]`` --
        ``blobs[w][k]``, the reference's order (``src/samplers.jl:238, :270``) -- or ``[k, nwalkers, m]`` sample-major."""
        if self.nblob <= 0:
            raise ValueError("this sampler's density returns no blobs (CDensity(..., nblob=m))")
        k = max(0, min(self.nsamples, (self.generation - self.cfg.nburnin) // self.cfg.nthin))
        out = np.empty((self.nlocal, k, self.nblob) if by_walker else (k, self.nlocal, self.nblob))
        _lib.check(self._L.kmc_sampler_get_blobs(self._h, None, _dp(out), 1 if by_walker else 0))
        return out

    def current_blobs(self):
        """The blob of every walker's current position, ``[nwalkers, m]`` (``blob0s``, ``src/samplers.jl:210, :264``)."""
        if self.nblob <= 0:
            raise ValueError("this sampler's density returns no blobs (CDensity(..., nblob=m))")
        out = np.empty((self.nrows, self.nblob))
        _lib.check(self._L.kmc_sampler_get_blobs(self._h, _dp(out), None, 0))
        return out

    def chain(self, logp: bool = True, by_walker: bool = False, out=None):
        """``(chain [nsamples_done, nlocal, ndim], chain_logp [nsamples_done, nlocal] | None)``; with ``by_walker`` in the
        reference's order, ``thetas[w][k]`` (``src/samplers.jl:219-221``): ``[nlocal, nsamples_done, ndim]`` and
        ``[nlocal, nsamples_done]``, transposed on the device (``kmc_sampler_get_chain_by_walker``; a streamed chain is
        reordered on the host).  ``out=(chain, chain_logp)``: C-contiguous float64 arrays of exactly those shapes to fill (by-walker
        read-out of a device chain) -- a caller that allocated and faulted them in while the device was sampling (``api.emcee``)."""
        ns = self.nsamples
        post = self.generation - self.cfg.nburnin
        done = 0 if post <= 0 else min(ns, post // self.cfg.nthin)
        if self._host_chain is not None or self._host_logp is not None:
            self.sync()                                   # KMC_STREAM_CHAIN: completes the copies of everything stored so far
            if self._host_by_walker:                       # streamed in the reference's order already
                ch = None if self._host_chain is None else self._host_chain[:, :done]
                lp = self._host_logp[:, :done] if (logp and self._host_logp is not None) else None
                if not by_walker:
                    ch = None if ch is None else np.ascontiguousarray(ch.transpose(1, 0, 2))
                    lp = None if lp is None else np.ascontiguousarray(lp.T)
                return ch, lp
            ch = None if self._host_chain is None else self._host_chain[:done]
            lp = self._host_logp[:done] if (logp and self._host_logp is not None) else None
            if by_walker:
                ch = None if ch is None else np.ascontiguousarray(ch.transpose(1, 0, 2))
                lp = None if lp is None else np.ascontiguousarray(lp.T)
            return ch, lp
        if by_walker:
            ch = lp = None
            if out is not None:
                ch, lp = out
                ok = (isinstance(ch, np.ndarray) and ch.dtype == np.float64 and ch.flags.c_contiguous and ch.shape == (self.nlocal, done, self.ndim) and
                      (not logp or (isinstance(lp, np.ndarray) and lp.dtype == np.float64 and lp.flags.c_contiguous and lp.shape == (self.nlocal, done))))
                if not ok:
                    raise ValueError("out: C-contiguous float64 arrays of shapes (nlocal, samples_done, ndim) and (nlocal, samples_done)")
            else:
                ch = np.empty((self.nlocal, done, self.ndim))
                lp = np.empty((self.nlocal, done)) if logp else None
            _lib.check(self._L.kmc_sampler_get_chain_by_walker(self._h, _dp(ch), _dp(lp) if logp else None))
            return ch, (lp if logp else None)
        ch = np.empty((done, self.nlocal, self.ndim))
        lp = np.empty((done, self.nlocal)) if logp else None
        _lib.check(self._L.kmc_sampler_get_chain(self._h, _dp(ch), _dp(lp) if logp else None))
        return ch, lp
