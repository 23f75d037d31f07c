"""Host-side mirror of the reference's public interface for the emcee path -- same names,
argument meaning and error behaviour -- over the C ABI:

* ``emcee(pdf, theta0s; niter, nburnin, nthin, a_scale, ...)``  reference ``src/samplers.jl:188-216``
* ``make_theta0s(theta0, ball_radius, pdf, nwalkers; ...)``      reference ``src/samplers.jl:311-349``
* ``squash_walkers(thetas, accept_ratio, logdensities, blobs; ...)``  reference ``src/samplers.jl:372-428``

Differences from the reference, all forced by the device boundary:
``pdf`` is a device density (:mod:`.densities`) or any host callable (evaluated on the host per
half-step; the only kind that can carry blobs, ``hasblob=True``); a ``seed`` keyword makes runs
reproducible (the reference never seeds); outputs are dense ndarrays indexed
``thetas[walker][sample]`` instead of vectors of vectors (blobs: a list per walker).
"""
from __future__ import annotations

import sys

import numpy as np

from . import _lib
from .densities import DeviceLogPdf, HostLogPdf
from .sampler import Sampler


def _fresh_seed() -> int:
    return int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).astype(np.uint64) @ np.array([1, 1 << 32], dtype=np.uint64))


def _free_device_bytes(device: int) -> float:
    """hipMemGetInfo through the library (no torch: importing it and creating its HIP context costs seconds and hundreds of MiB)."""
    import ctypes as C
    free, total = C.c_uint64(0), C.c_uint64(0)
    _lib.check(_lib.lib().kmc_device_free_bytes(int(device), C.byref(free), C.byref(total)))
    return float(free.value)


def emcee_counts(niter: int, nwalkers: int, nburnin=None, nthin: int = 1):
    """The reference's integer bookkeeping: ``niter``/``nburnin`` count log-pdf evaluations over
    all walkers (``src/samplers.jl:159``); per-walker counts are the floor divisions of
    ``src/samplers.jl:203-204`` and the stored samples per walker that of ``:234``."""
    if nburnin is None:
        nburnin = niter // 2                                   # :190
    niter_walker = niter // nwalkers                           # :203
    nburnin_walker = nburnin // nwalkers                       # :204
    nsamples_walker = (niter_walker - nburnin_walker) // nthin  # :234
    return niter_walker, nburnin_walker, nsamples_walker


def emcee(pdf, theta0s, niter: int = 10 ** 5, nburnin=None, nthin: int = 1, a_scale: float = 2.0,
          use_progress_meter: bool = True, hasblob: bool = False, init_blobs=None, reduce_blob=None,
          seed=None, device: int = 0, dtype: str = "f64", stream_chain=None):
    """The affine-invariant ensemble sampler, on one MI355X.  ``dtype="f32"`` keeps the walkers in single
    precision on the device (a throughput option, device densities; everything returned is still float64).

    Returns ``(thetas, accept_ratio, logdensities, blobs)`` like the reference
    (``src/samplers.jl:292``): ``thetas[w][k]`` is sample ``k`` of walker ``w``
    (shape ``[nwalkers, nsamples]`` for scalar walkers, ``[nwalkers, nsamples, ndim]`` otherwise),
    ``accept_ratio[w]``, ``logdensities[w][k]``, and ``blobs`` (``None`` unless ``hasblob``).

    ``stream_chain``: ``True`` streams the stored samples to host memory block by block while sampling (the device
    keeps a small ring; the chain is bounded by host RAM instead of HBM, like the reference's growing vectors,
    ``:268-272``), ``False`` keeps the whole chain on the device until the end, ``None`` (default) streams when the
    chain would not fit the device (more than 70 % of its free memory; page-locking the host arrays costs about as much
    as downloading a chain that does fit).

    ``hasblob=True`` (``:150-151, :194-196``): ``pdf`` is a host callable returning ``(p, blob)``; the blobs
    stay on the host and follow the device's accept decisions.  ``blobs[w] = init_blobs(blob0s[w],
    nsamples_walker)`` (default: an empty list) and ``reduce_blob(blobs[w], blob)`` (default: append) is
    called for every stored sample with the walker's current blob (``:270``).  Or ``pdf`` is a ``CDensity(..., nblob=m)``:
    the blob is then ``m`` doubles computed on the device with the log-pdf, carried next to it by the kernels, and ``blobs``
    comes back as an array ``[nwalkers, nsamples, m]`` (``blobs[w][k]``, the default ``push!`` reduction) -- or, with
    ``init_blobs`` / ``reduce_blob`` given, whatever they build when fed each walker's stored series in order.
    """
    theta0s = np.array(theta0s, dtype=np.float64)              # :198 deepcopy
    scalar_walkers = theta0s.ndim == 1
    # walkers of any array shape (the reference asks of theta0s[1] only `.+`, `.*` and `length`, :156 -- a Matrix qualifies):
    # the sampler sees them flattened (N = length(theta0s[1]), :243), a host callable and the caller see the shape
    walker_shape = theta0s.shape[1:] if theta0s.ndim > 2 else None
    if walker_shape is not None:
        theta0s = theta0s.reshape(theta0s.shape[0], -1)
        if callable(pdf) and not isinstance(pdf, DeviceLogPdf):
            fn_shaped = pdf
            pdf = lambda x: fn_shaped(np.asarray(x).reshape(walker_shape))
    if not hasblob and (init_blobs is not None or reduce_blob is not None):
        raise ValueError("init_blobs / reduce_blob need hasblob=True")
    if hasblob:
        if isinstance(pdf, HostLogPdf):
            if not pdf.hasblob:
                raise ValueError("hasblob=True needs HostLogPdf(..., hasblob=True)")
        elif isinstance(pdf, DeviceLogPdf):
            if int(getattr(pdf, "nblob", 0) or 0) <= 0:
                raise NotImplementedError("hasblob=True needs a pdf that returns a blob: a host callable returning (p, blob), or a "
                                          "CDensity(..., nblob=m) whose body fills blob[0..m) on the device (menu and term / pair "
                                          "densities return the log-pdf alone)")
        elif callable(pdf):
            pdf = HostLogPdf(pdf, scalar=scalar_walkers, hasblob=True)
    if not isinstance(pdf, DeviceLogPdf):
        if not callable(pdf):
            raise TypeError(f"pdf must be a log-density object or a callable; got {type(pdf).__name__}")
        # an arbitrary closure, as in the reference (:257): the moves and the accept test run on the
        # device, the closure is evaluated on the host on each half-step's batch of proposals
        pdf = HostLogPdf(pdf, scalar=scalar_walkers)
    if scalar_walkers:
        theta0s = theta0s[:, None]
    if theta0s.ndim != 2:
        raise ValueError("theta0s must be a vector of walkers (scalars or equal-length vectors)")
    nwalkers, ndim = theta0s.shape
    # the reference's asserts, in its order (:200, :202, :205)
    if not a_scale > 1:
        raise AssertionError("a_scale>1")
    if nwalkers % 2 != 0:
        raise AssertionError("Use an even number of walkers.")
    niter_walker, nburnin_walker, nsamples_walker = emcee_counts(niter, nwalkers, nburnin, nthin)
    if nwalkers < ndim + 2:
        raise AssertionError("Use more walkers: at least DOF+2, but better many more.")
    if seed is None:
        seed = _fresh_seed()

    device_blobs = bool(hasblob and not isinstance(pdf, HostLogPdf))     # a CDensity(..., nblob=m): blobs computed and carried on the device
    if device_blobs and stream_chain:
        raise NotImplementedError("device blobs with a streamed chain: not supported (thin the chain, or stream_chain=False)")
    if device_blobs:
        stream_chain = False
    if stream_chain is None:
        # a chain that would not fit the device is streamed to host memory while sampling; the device is only asked when the
        # chain is large at all (> 1 GiB)
        chain_bytes = nsamples_walker * nwalkers * (ndim + 1 + ndim % 2) * 8
        stream_chain = dtype == "f64" and chain_bytes > (1 << 30) and chain_bytes > 0.7 * _free_device_bytes(device)
    # (host arrays that cannot be page-locked are no reason to fail: the library then stages the by-walker blocks itself)
    with Sampler(pdf, nwalkers, ndim, niter_walker, nburnin_walker, nthin, a_scale, seed, store_chain=True, store_logp=True,
                 device=device, dtype=dtype, stream_chain=bool(stream_chain), chain_by_walker=True, store_blobs=device_blobs) as s:
        try:
            s.set_positions(theta0s)
        except _lib.KmcError as e:
            if e.status == _lib.ERR_NONFINITE_LOGP:
                raise ValueError(f"{e} (use make_theta0s to build an initial ensemble with pdf > -Inf)") from e
            raise
        blobs = None
        if device_blobs:
            blob0s = s.current_blobs()                             # :209-210, from the initial evaluations (on the device)
        elif hasblob:
            if init_blobs is None:
                init_blobs = lambda blob0, nsamples: []        # init_output_vector :80-85
            if reduce_blob is None:
                reduce_blob = lambda bs, b: bs.append(b)       # push!  :196
            blob0s = list(pdf.last_blobs)                      # :209-210, from the initial evaluations
            blobs = [init_blobs(blob0s[w], nsamples_walker) for w in range(nwalkers)]   # :238

            def on_accepted(accepted, row0, generation, stored):
                batch = pdf.last_blobs
                for i in np.nonzero(accepted)[0]:
                    blob0s[row0 + i] = batch[i]                # :264
                if stored:
                    for w in range(row0, row0 + accepted.shape[0]):
                        reduce_blob(blobs[w], blob0s[w])       # :270

            pdf.on_accepted = on_accepted
        # the dense output arrays (:219-221) exist before the run and are faulted in by a helper thread while the device samples: a fresh
        # allocation's pages are otherwise created one fault at a time by the read-out's copy threads (4 096 x 4, 1 000 samples per walker:
        # 15.6 ms of read-out after 6.4 ms of sampling)
        out_arrays = warm = None
        if not stream_chain and nsamples_walker > 0 and nwalkers * nsamples_walker * (ndim + 1) * 8 >= (8 << 20):
            import threading
            out_arrays = (np.empty((nwalkers, nsamples_walker, ndim)), np.empty((nwalkers, nsamples_walker)))
            L = _lib.lib()
            warm = threading.Thread(target=lambda: [L.kmc_host_prefault(a.ctypes.data, a.nbytes, 4) for a in out_arrays], daemon=True)
            warm.start()
        try:
            _run_generations(s, niter, nwalkers, niter_walker, nburnin_walker, use_progress_meter)
            if warm is not None:
                warm.join()
            thetas, logdensities = s.chain(logp=True, by_walker=True, out=out_arrays)     # [walker][sample][dim], [walker][sample]: :219-221
            accept_ratio = s.accept_ratio()
            if device_blobs:
                series = s.blobs(by_walker=True)               # [walker][sample][m]: the walker's current blob at every stored sample (:270)
                if init_blobs is None and reduce_blob is None:
                    blobs = series                             # push! into an empty vector, densely
                else:                                          # the caller's reduction, fed the same series in the same order
                    ib = init_blobs if init_blobs is not None else (lambda blob0, nsamples: [])
                    rb = reduce_blob if reduce_blob is not None else (lambda bs, b: bs.append(b))
                    blobs = [ib(blob0s[w], nsamples_walker) for w in range(nwalkers)]        # :238
                    for w in range(nwalkers):
                        for k in range(series.shape[1]):
                            rb(blobs[w], series[w, k])
        finally:
            if hasblob and not device_blobs:
                pdf.on_accepted = None                         # the closure holds this call's blob storage

    if scalar_walkers:
        thetas = thetas[:, :, 0]
    assert thetas.shape[1] == nsamples_walker
    if walker_shape is not None:
        thetas = thetas.reshape(thetas.shape[:2] + tuple(walker_shape))
    return thetas, accept_ratio, logdensities, blobs


def _run_generations(s, niter, nwalkers, niter_walker, nburnin_walker, use_progress_meter):
    """All generations of one emcee call, in up to 20 pieces with a progress line on stderr after each
    (the reference's ProgressMeter values, ``src/samplers.jl:275-284``) or in one piece."""
    if use_progress_meter and niter_walker > 0:
        # (a long job -- >= 4096 generations -- in pieces of >= 1024: a sampler measures its launch modes in a run of >= 896 generations, kissmcmc_hip.h)
        nchunks = min(20, niter_walker if niter_walker < 4096 else niter_walker // 1024)
        done = 0
        for c in range(nchunks):
            upto = (niter_walker * (c + 1)) // nchunks
            s.run(upto - done)
            done = upto
            s.sync()
            na = s.naccept().astype(np.float64)
            nn = max(1, done - nburnin_walker if done > nburnin_walker else done)
            macc, sacc = na.mean(), np.sqrt(na.var(ddof=1))          # :276-278
            outl = int(np.sum(np.abs(na - macc) > 2 * sacc))
            print(f"\remcee, niter={niter}, nwalkers={nwalkers}: generation {done}/{niter_walker} "
                  f"accept_ratio_mean={macc / nn:.3g} accept_ratio_std={sacc / nn:.3g} "
                  f"accept_ratio_outliers={outl} burnin_phase={done <= nburnin_walker}",      # :279-283
                  end="", file=sys.stderr)
        print(file=sys.stderr)
    else:
        s.run(niter_walker)
    s.sync()


def make_theta0s(theta0, ball_radius, pdf, nwalkers: int, ball_radius_halfing_steps: int = 7,
                 ntries: int = 100, hasblob: bool = False, rng=None):
    """Initial ensemble in a Gaussian ball around ``theta0`` with ``pdf > -Inf``
    (reference ``src/samplers.jl:311-349``).  ``rng`` is a ``numpy.random.Generator`` (or a seed);
    the normal draws are consumed walker by walker, try by try, like the reference's ``randn``.

    Returns an array of shape ``[nwalkers]`` (scalar ``theta0``) or ``[nwalkers, npara]``.
    Unlike the reference -- whose final ``error(...)`` is unreachable -- this
    raises when no admissible point is found for a walker.
    """
    if not callable(pdf):
        raise TypeError("pdf must be callable")
    if hasblob and isinstance(pdf, DeviceLogPdf) and not getattr(pdf, "hasblob", False):
        if int(getattr(pdf, "nblob", 0) or 0) <= 0:
            raise NotImplementedError("hasblob=True needs a host callable returning (p, blob), or a CDensity(..., nblob=m)")
        hasblob = False                                        # a device density with blobs: calling it returns pdf(tmp)[1] alone (:336)
    if hasblob and not getattr(pdf, "hasblob", False):          # :333-337  p0, blob0 = pdf(tmp)
        pdf_blob = pdf
        pdf = lambda t: pdf_blob(t)[0]
    rng = np.random.default_rng(rng)
    scalar = np.ndim(theta0) == 0
    theta0 = float(theta0) if scalar else np.asarray(theta0, dtype=np.float64)
    npara = 1 if scalar else theta0.shape[0]                   # :315
    if np.ndim(ball_radius) == 0 and not scalar:               # :316-318
        ball_radius = np.ones(npara) * float(ball_radius)
    ball_radius = float(ball_radius) if np.ndim(ball_radius) == 0 else np.asarray(ball_radius, dtype=np.float64)
    if np.size(ball_radius) != npara:                          # :319
        raise AssertionError("length(ball_radius)==npara")

    # Fast path: when every first try is admissible the result equals the sequential loop's.
    state = rng.bit_generator.state
    if scalar:
        cand = theta0 + rng.standard_normal(nwalkers) * ball_radius
        rows = cand[:, None]
    else:
        cand = theta0 + rng.standard_normal((nwalkers, npara)) * ball_radius
        rows = cand
    finite = getattr(pdf, "finite_rows", None)
    ok = finite(rows) if finite is not None else np.array([pdf(r if not scalar else r[0]) > -np.inf for r in rows])
    if bool(np.all(ok)):
        return cand
    rng.bit_generator.state = state

    out = []
    for i in range(1, nwalkers + 1):                           # :323
        for k in range(1, ball_radius_halfing_steps + 1):      # :324
            ball_radius = ball_radius * (1 / 2 ** (k - 1))     # :326 (never reset, as in the reference)
            for _ in range(ntries):                            # :327
                if npara == 1:                                 # :328-332
                    tmp = theta0 + rng.standard_normal() * ball_radius
                else:
                    tmp = theta0 + rng.standard_normal(npara) * ball_radius
                if pdf(tmp) > -np.inf:                         # :336-338
                    out.append(tmp)
                    break
            if len(out) == i:                                  # :343
                break
        if len(out) != i:
            raise RuntimeError("Could not find suitable initial theta.  PDF is zero in too many places inside ball.")
    return np.array(out, dtype=np.float64).reshape((nwalkers,) if scalar else (nwalkers, npara))


def squash_walkers(thetas, accept_ratio, logdensities=None, blobs=None, drop_low_accept_ratio: bool = False,
                   drop_fact=2, verbose: bool = True, order: bool = False, merge_blobs=None):
    """Put the samples of all walkers into one array (reference ``src/samplers.jl:372-428``).

    Default order is walker-major (``[w1 samples..., w2 samples..., ...]``, ``:398-399``);
    ``order=True`` gives sample-major with walkers in index order inside each step (``:415-426``).
    Returns ``(thetas, mean(accept_ratio[kept]), logdensities, blobs)``.
    """
    thetas = np.asarray(thetas)
    accept_ratio = np.asarray(accept_ratio, dtype=np.float64)
    nwalkers = accept_ratio.shape[0]                           # :379
    if drop_low_accept_ratio:                                  # :380-393
        ma = float(np.median(accept_ratio))
        sa = float(np.std(accept_ratio, ddof=1))
        if verbose:
            print(f"Median accept ratio is {ma}, standard deviation is {sa}\n")
        keep = ~(accept_ratio <= ma - drop_fact * sa)
        if verbose:
            for nc in np.nonzero(~keep)[0]:
                print(f"Dropping walker {nc + 1} with low accept ratio {accept_ratio[nc]}")
        walkers2keep = np.nonzero(keep)[0]
    else:
        walkers2keep = np.arange(nwalkers)                     # :395

    def flat(a):
        a = np.asarray(a)
        if len(walkers2keep) != a.shape[0]:
            a = a[walkers2keep]                                # [kept][sample](...); keeping all: a view, no copy
        if order:                                              # :415-426: stable sort by sample index
            a = np.swapaxes(a, 0, 1)
        return np.ascontiguousarray(a).reshape((-1,) + a.shape[2:])

    t = flat(thetas)                                           # :398-399
    l = None if logdensities is None else flat(logdensities)   # :401-406
    b = None
    if isinstance(blobs, np.ndarray) and blobs.ndim >= 2 and merge_blobs is None:
        b = flat(blobs)                                        # a dense series [walker][sample](m) (device blobs): append! = concatenation
    elif blobs is not None:                                    # :408-413
        import copy
        if merge_blobs is None:
            merge_blobs = lambda b1, b2: b1.extend(b2)         # append!
        b = copy.deepcopy(blobs[walkers2keep[0]])
        for w in walkers2keep[1:]:
            merge_blobs(b, blobs[w])
        if order:                                              # :415-421: b[perm], walker-major -> sample-major
            nc, ns = len(walkers2keep), thetas.shape[1]
            if len(b) != nc * ns:
                raise ValueError("order=True needs one stored blob per sample (blobs kept in vectors)")
            b = [b[w * ns + k] for k in range(ns) for w in range(nc)]
    return t, float(np.mean(accept_ratio[walkers2keep])), l, b   # :427
