"""Host-side mirror of the reference's ``metropolis`` (reference ``src/samplers.jl:9-128``) over the C ABI,
plus its data-parallel form, many independent chains at once.

* ``metropolis(pdf, sample_ppdf, theta0; niter, nburnin, nthin, ...)`` -- the reference's signature and
  return shapes, one chain (a single GPU lane: a drop-in, not a fast path);
* ``metropolis_chains(pdf, sample_ppdf, theta0s; ...)`` -- one chain per row of ``theta0s``, one chain per
  lane; returns the same ``[chain][sample]`` containers as ``emcee`` so ``squash_walkers`` applies.

Fast path: ``sample_ppdf`` a :class:`GaussianStep` -- the symmetric jump every reference test uses,
``theta -> c .* randn(n) .+ theta`` (``test/runtests.jl:54,59,64,75``) -- and ``pdf`` a menu or runtime-compiled
density: the whole chain runs inside one kernel, one chain per lane.

General path (the reference takes ANY two closures, ``src/samplers.jl:59-61``; its README-style call
``metropolis(pdf, sample_prop_normal, theta0)`` works as it is): a callable ``pdf`` and / or a callable ``sample_ppdf``
stay on the host and are called once per iteration on the batch of all chains, while the accept test (``:101``), the
counters and the sample storage stay on the device -- launch- and PCIe-bound (tens of microseconds per iteration), the
same trade as ``emcee``'s host-evaluated densities.  ``hasblob=True`` needs a callable ``pdf`` returning ``(p, blob)``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .densities import DeviceLogPdf, HostLogPdf


class GaussianStep:
    """Symmetric proposal ``theta1 = theta0 .+ scale .* randn(ndim)`` (``scale``: scalar or one per dimension)."""

    def __init__(self, scale=1.0):
        self.scale = np.atleast_1d(np.asarray(scale, dtype=np.float64))
        if self.scale.ndim != 1 or not np.all(np.isfinite(self.scale)):
            raise ValueError("scale must be a finite scalar or vector")

    def scales(self, ndim: int) -> np.ndarray:
        if self.scale.size not in (1, ndim):
            raise ValueError(f"GaussianStep has {self.scale.size} scales for {ndim} dimensions")
        return np.ascontiguousarray(np.broadcast_to(self.scale, (ndim,)), dtype=np.float64)

    def __repr__(self):
        return f"GaussianStep({self.scale.tolist() if self.scale.size > 1 else float(self.scale[0])})"


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class HostProposal:
    """ANY callable as ``sample_ppdf`` (``src/samplers.jl:41, :98``: a symmetric proposal ``theta0 -> theta1``), kept on
    the host and called per iteration on every chain's current state (a float when ``scalar`` else a 1-D array; with
    ``vectorized=True`` on the whole batch ``[nchains, ndim]`` at once)."""

    def __init__(self, fn, scalar: bool = False, vectorized: bool = False):
        if not callable(fn):
            raise TypeError("sample_ppdf must be callable")
        self.fn, self.scalar, self.vectorized = fn, bool(scalar), bool(vectorized)
        self.error = None

        def _cb(rows, nrows, ndim, out, _user):
            try:
                X = np.ctypeslib.as_array(rows, shape=(nrows, ndim))
                Y = np.ctypeslib.as_array(out, shape=(nrows, ndim))
                if self.vectorized:
                    Y[:] = np.asarray(self.fn(X[:, 0] if self.scalar else X), dtype=np.float64).reshape(nrows, ndim)
                else:
                    for i in range(nrows):
                        Y[i] = self.fn(float(X[i, 0])) if self.scalar else np.asarray(self.fn(X[i].copy()), dtype=np.float64)
                return 0
            except BaseException as e:      # never unwind through the C frames
                self.error = e
                return 1

        self.c_callback = _lib.HOST_PROPOSE_FN(_cb)


def run_chains(pdf, sample_ppdf, theta0s, niter, nburnin, nthin, seed, device=0, store_chain=True, store_logp=True,
               moments=False, scalar=False, by_chain=False, store_blobs=False):
    """``kmc_metropolis_run`` on dense arrays.  Returns a dict: ``chain [nsamples, nchains, ndim]``,
    ``chain_logp [nsamples, nchains]``, ``accept_ratio``, ``naccept``, ``final_pos``, ``final_logp``,
    ``chain_sum``/``chain_sumsq [nchains, ndim]``, ``nsamples``, ``device_ms``; with ``store_blobs`` (a ``CDensity(..., nblob=m)``)
    also ``blobs [nsamples, nchains, m]`` (``[nchains, nsamples, m]`` with ``by_chain``) and ``final_blob [nchains, m]``.

    ``pdf``: a device density, a :class:`~.densities.HostLogPdf` or any callable (wrapped in one); ``sample_ppdf``: a
    :class:`GaussianStep`, a :class:`HostProposal` or any callable (wrapped in one).  ``scalar``: bare callables get a
    float instead of a 1-element array (the reference's convention for a scalar ``theta0``)."""
    if not isinstance(pdf, DeviceLogPdf):
        if not callable(pdf):
            raise TypeError(f"pdf must be a log-density object or a callable; got {type(pdf).__name__}")
        pdf = HostLogPdf(pdf, scalar=scalar)             # an arbitrary closure, as in the reference (:99): evaluated on the host
    if not isinstance(sample_ppdf, (GaussianStep, HostProposal)):
        if not callable(sample_ppdf):
            raise TypeError("sample_ppdf must be a GaussianStep (theta -> scale .* randn(n) .+ theta) or a callable")
        sample_ppdf = HostProposal(sample_ppdf, scalar=scalar)
    theta0s = np.ascontiguousarray(np.asarray(theta0s, dtype=np.float64))
    if theta0s.ndim != 2:
        raise ValueError("theta0s must be [nchains, ndim]")
    nchains, ndim = theta0s.shape
    pdf.check_ndim(ndim)
    c = _lib.MetropolisConfig()
    c.dtype = _lib.F64
    c.density = pdf.density_id
    p = list(pdf.params()) + [0.0] * 8
    for i in range(8):
        c.params[i] = float(p[i])
    c.nchains, c.ndim, c.niter, c.nburnin, c.nthin = nchains, ndim, int(niter), int(nburnin), int(nthin)
    step = None
    if isinstance(sample_ppdf, GaussianStep):
        step = sample_ppdf.scales(ndim)
        c.step = _dp(step)
    else:
        c.host_propose = C.cast(sample_ppdf.c_callback, C.c_void_p)
    c.seed = int(seed)
    c.flags = (_lib.STORE_CHAIN if store_chain else 0) | (_lib.STORE_LOGP if store_logp else 0) | (_lib.MOMENTS if moments else 0)
    if by_chain:                    # chain [nchains, nsamples, ndim], chain_logp [nchains, nsamples]: reordered on the device
        c.flags |= _lib.CHAIN_BY_WALKER
    nblob = int(getattr(pdf, "nblob", 0) or 0)
    if store_blobs:
        if nblob <= 0:
            raise ValueError("store_blobs needs a CDensity(..., nblob=m)")
        c.flags |= _lib.STORE_BLOBS
    c.device = int(device)
    c.user_density = pdf.user_handle
    if isinstance(pdf, HostLogPdf):
        c.host_logpdf = C.cast(pdf.c_callback, C.c_void_p)
        if pdf.c_accepted is not None:
            c.host_accepted = C.cast(pdf.c_accepted, C.c_void_p)
    L = _lib.lib()
    _lib.check(L.kmc_metropolis_validate(C.byref(c)))
    ns = max(0, (int(niter) - int(nburnin)) // int(nthin)) if niter > nburnin else 0     # src/samplers.jl:88
    chain = np.zeros((nchains, ns, ndim) if by_chain else (ns, nchains, ndim)) if store_chain else None
    chain_logp = np.zeros((nchains, ns) if by_chain else (ns, nchains)) if store_logp else None
    acc = np.zeros(nchains)
    nacc = np.zeros(nchains, dtype=np.int64)
    fpos = np.zeros((nchains, ndim))
    flogp = np.zeros(nchains)
    csum = np.zeros((nchains, ndim)) if moments else None
    csq = np.zeros((nchains, ndim)) if moments else None
    o = _lib.MetropolisOutputs()
    o.chain, o.chain_logp, o.accept_ratio = _dp(chain), _dp(chain_logp), _dp(acc)
    o.naccept = nacc.ctypes.data_as(C.POINTER(C.c_int64))
    o.final_pos, o.final_logp, o.chain_sum, o.chain_sumsq = _dp(fpos), _dp(flogp), _dp(csum), _dp(csq)
    blobs = np.zeros((nchains, ns, nblob) if by_chain else (ns, nchains, nblob)) if store_blobs else None
    fblob = np.zeros((nchains, nblob)) if store_blobs else None
    o.blobs, o.final_blob = _dp(blobs), _dp(fblob)
    with np.errstate(all="ignore"):
        status = L.kmc_metropolis_run(C.byref(c), _dp(theta0s), C.byref(o))
    for obj in (pdf, sample_ppdf):                        # an exception raised inside a host callback: re-raise it here
        err = getattr(obj, "error", None)
        if status != _lib.OK and err is not None:
            obj.error = None
            raise err
    _lib.check(status)
    assert o.nsamples == ns
    return dict(chain=chain, chain_logp=chain_logp, accept_ratio=acc, naccept=nacc, final_pos=fpos, final_logp=flogp,
                chain_sum=csum, chain_sumsq=csq, nsamples=ns, device_ms=o.device_ms, blobs=blobs, final_blob=fblob)


def _fresh_seed() -> int:
    return int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).astype(np.uint64) @ np.array([1, 1 << 32], dtype=np.uint64))


def _blob_plumbing(pdf, hasblob, init_blobs, reduce_blob, scalar, nchains, nsamples):
    """The reference's blob handling (``src/samplers.jl:70-72, :100-103, :116-118``) on the host: ``pdf`` returns ``(p, blob)``;
    ``blob0`` follows the device's accept decisions, ``reduce_blob(blobs, blob0)`` runs for every stored sample."""
    if not hasblob:
        if init_blobs is not None or reduce_blob is not None:
            raise ValueError("init_blobs / reduce_blob need hasblob=True")
        return pdf, None
    if isinstance(pdf, HostLogPdf):
        if not pdf.hasblob:
            raise ValueError("hasblob=True needs HostLogPdf(..., hasblob=True)")
    elif isinstance(pdf, DeviceLogPdf):
        raise NotImplementedError("hasblob=True needs a pdf that returns a blob: a host callable returning (p, blob), or a CDensity(..., nblob=m)")
    else:
        pdf = HostLogPdf(pdf, scalar=scalar, hasblob=True)
    if init_blobs is None:
        init_blobs = lambda blob0, ns: []                 # init_output_vector :80-85
    if reduce_blob is None:
        reduce_blob = lambda bs, b: bs.append(b)          # push!  :67
    state = dict(blob0s=None, blobs=None)

    def on_accepted(accepted, row0, iteration, stored):
        if state["blob0s"] is None:                       # first call: the blobs of the initial evaluation (:70) were replaced by
            raise RuntimeError("initial blobs missing")   # the first batch's -- they are captured before the run (below)
        batch = pdf.last_blobs
        for i in np.nonzero(accepted)[0]:
            state["blob0s"][i] = batch[i]                 # :103
        if stored:
            for w in range(nchains):
                reduce_blob(state["blobs"][w], state["blob0s"][w])      # :117
    pdf.on_accepted = on_accepted
    return pdf, (state, init_blobs, nsamples)


def metropolis(pdf, sample_ppdf, theta0, niter: int = 10 ** 5, nburnin=None, nthin: int = 1,
               use_progress_meter: bool = True, hasblob: bool = False, init_blobs=None, reduce_blob=None,
               seed=None, device: int = 0):
    """One Metropolis chain with the reference's signature (``src/samplers.jl:59-77``); ``pdf`` and ``sample_ppdf`` may be
    device objects (fast: the chain runs in one kernel) or ANY callables, as in the reference (host route).

    Returns ``(thetas, accept_ratio, logdensities, blobs)`` (``:128``): ``thetas[k]`` is stored sample ``k``
    (shape ``[nsamples]`` for a scalar ``theta0``, ``[nsamples, ndim]`` for a vector), ``accept_ratio`` a float,
    ``blobs`` ``None`` unless ``hasblob``.  ``use_progress_meter`` is accepted and ignored.
    """
    if nburnin is None:
        nburnin = niter // 2                                              # :63
    scalar = np.ndim(theta0) == 0
    th = np.array(theta0, dtype=np.float64).reshape(1, -1)                # :68 deepcopy
    if seed is None:
        seed = _fresh_seed()
    thetas, acc, logd, blobs = _run(pdf, sample_ppdf, th, niter, nburnin, nthin, hasblob, init_blobs, reduce_blob, seed, device, scalar)
    return (thetas[0], float(acc[0]), logd[0], None if blobs is None else blobs[0])


def _run(pdf, sample_ppdf, th, niter, nburnin, nthin, hasblob, init_blobs, reduce_blob, seed, device, scalar):
    nchains = th.shape[0]
    ns = max(0, (int(niter) - int(nburnin)) // int(nthin)) if niter > nburnin else 0
    if hasblob and isinstance(pdf, DeviceLogPdf) and not isinstance(pdf, HostLogPdf) and int(getattr(pdf, "nblob", 0) or 0) > 0:
        # a CDensity(..., nblob=m): blob0 follows the chain on the device (:72, :103), the blob of every stored sample comes back
        # (:117); a caller's init_blobs / reduce_blob are then fed each chain's series in order
        if not isinstance(sample_ppdf, GaussianStep):
            raise NotImplementedError("device blobs run in the in-kernel chains: sample_ppdf must be a GaussianStep")
        r = run_chains(pdf, sample_ppdf, th, niter, nburnin, nthin, seed, device, scalar=scalar, by_chain=True, store_blobs=True)
        series = r["blobs"]
        if init_blobs is None and reduce_blob is None:
            blobs = series
        else:
            ib = init_blobs if init_blobs is not None else (lambda blob0, n: [])
            rb = reduce_blob if reduce_blob is not None else (lambda bs, b: bs.append(b))
            _, blob0s = pdf.eval_with_blobs(th)                                # p0, blob0 = pdf(theta0)  :70
            blobs = [ib(blob0s[w], ns) for w in range(nchains)]               # :90
            for w in range(nchains):
                for k in range(series.shape[1]):
                    rb(blobs[w], series[w, k])                                 # :117
        thetas = r["chain"][:, :, 0] if scalar else r["chain"]
        return thetas, r["accept_ratio"], r["chain_logp"], blobs
    pdf, blobctx = _blob_plumbing(pdf, hasblob, init_blobs, reduce_blob, scalar, nchains, ns)
    try:
        if blobctx is not None:
            state, init_blobs, _ = blobctx
            pdf.eval_rows(th)                                             # p0, blob0 = pdf(theta0)  :70 (the library evaluates it again)
            state["blob0s"] = list(pdf.last_blobs)
            state["blobs"] = [init_blobs(state["blob0s"][w], ns) for w in range(nchains)]       # :90
        r = run_chains(pdf, sample_ppdf, th, niter, nburnin, nthin, seed, device, scalar=scalar, by_chain=True)
    finally:
        if blobctx is not None:
            pdf.on_accepted = None
    thetas = r["chain"]                                         # [chain][sample][dim], :113
    if scalar:
        thetas = thetas[:, :, 0]
    return thetas, r["accept_ratio"], r["chain_logp"], (None if blobctx is None else blobctx[0]["blobs"])


def metropolis_chains(pdf, sample_ppdf, theta0s, niter: int = 10 ** 5, nburnin=None, nthin: int = 1,
                      hasblob: bool = False, init_blobs=None, reduce_blob=None, seed=None, device: int = 0):
    """Many independent Metropolis chains, one per row of ``theta0s`` (``[nchains]`` scalars or
    ``[nchains, ndim]``); ``niter``/``nburnin`` count steps PER CHAIN, as in ``metropolis``.

    Returns ``(thetas, accept_ratio, logdensities, blobs)`` shaped like ``emcee``'s output
    (``thetas[chain][sample]``), so ``squash_walkers(*result)`` concatenates the chains.
    """
    if nburnin is None:
        nburnin = niter // 2
    theta0s = np.array(theta0s, dtype=np.float64)
    scalar = theta0s.ndim == 1
    if scalar:
        theta0s = theta0s[:, None]
    if seed is None:
        seed = _fresh_seed()
    return _run(pdf, sample_ppdf, theta0s, niter, nburnin, nthin, hasblob, init_blobs, reduce_blob, seed, device, scalar)
