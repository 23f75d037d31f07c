"""Host-side mirror of the reference's ``metropolis`` (reference ``src/samplers.jl:9-128``) over the C ABI,
plus its data-parallel form, many independent chains at once.

* ``metropolis(pdf, sample_ppdf, theta0; niter, nburnin, nthin, ...)`` -- the reference's signature and
  return shapes, one chain (a single GPU lane: a drop-in, not a fast path);
* ``metropolis_chains(pdf, sample_ppdf, theta0s; ...)`` -- one chain per row of ``theta0s``, one chain per
  lane; returns the same ``[chain][sample]`` containers as ``emcee`` so ``squash_walkers`` applies.

On the device ``sample_ppdf`` is a :class:`GaussianStep` -- the symmetric jump every reference test uses,
``theta -> c .* randn(n) .+ theta`` (``test/runtests.jl:54,59,64,75``) -- and ``pdf`` a menu or
runtime-compiled density.  An arbitrary closure for either cannot run inside the kernel and is refused.
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


def run_chains(pdf, sample_ppdf, theta0s, niter, nburnin, nthin, seed, device=0, store_chain=True, store_logp=True,
               moments=False):
    """``kmc_metropolis_run`` on dense arrays.  Returns a dict: ``chain [nsamples, nchains, ndim]``,
    ``chain_logp [nsamples, nchains]``, ``accept_ratio``, ``naccept``, ``final_pos``, ``final_logp``,
    ``chain_sum``/``chain_sumsq [nchains, ndim]``, ``nsamples``, ``device_ms``."""
    if isinstance(pdf, HostLogPdf) or not isinstance(pdf, DeviceLogPdf):
        raise TypeError("the many-chain Metropolis kernel evaluates pdf on the device: pass a menu density "
                        "(GaussianIso, Exponential, Rosenbrock, LogNormal, MvNormal2) or an ExprDensity")
    if not isinstance(sample_ppdf, GaussianStep):
        raise TypeError("sample_ppdf must be a GaussianStep (theta -> scale .* randn(n) .+ theta): an arbitrary "
                        "proposal closure cannot run inside the kernel")
    theta0s = np.ascontiguousarray(np.asarray(theta0s, dtype=np.float64))
    if theta0s.ndim != 2:
        raise ValueError("theta0s must be [nchains, ndim]")
    nchains, ndim = theta0s.shape
    pdf.check_ndim(ndim)
    step = sample_ppdf.scales(ndim)
    c = _lib.MetropolisConfig()
    c.dtype = _lib.F64
    c.density = pdf.density_id
    p = list(pdf.params()) + [0.0] * 8
    for i in range(8):
        c.params[i] = float(p[i])
    c.nchains, c.ndim, c.niter, c.nburnin, c.nthin = nchains, ndim, int(niter), int(nburnin), int(nthin)
    c.step = _dp(step)
    c.seed = int(seed)
    c.flags = (_lib.STORE_CHAIN if store_chain else 0) | (_lib.STORE_LOGP if store_logp else 0) | (_lib.MOMENTS if moments else 0)
    c.device = int(device)
    c.user_density = pdf.user_handle
    L = _lib.lib()
    _lib.check(L.kmc_metropolis_validate(C.byref(c)))
    ns = max(0, (int(niter) - int(nburnin)) // int(nthin)) if niter > nburnin else 0     # src/samplers.jl:88
    chain = np.zeros((ns, nchains, ndim)) if store_chain else None
    chain_logp = np.zeros((ns, nchains)) if store_logp else None
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
    with np.errstate(all="ignore"):
        _lib.check(L.kmc_metropolis_run(C.byref(c), _dp(theta0s), C.byref(o)))
    assert o.nsamples == ns
    return dict(chain=chain, chain_logp=chain_logp, accept_ratio=acc, naccept=nacc, final_pos=fpos, final_logp=flogp,
                chain_sum=csum, chain_sumsq=csq, nsamples=ns, device_ms=o.device_ms)


def _fresh_seed() -> int:
    return int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).astype(np.uint64) @ np.array([1, 1 << 32], dtype=np.uint64))


def _no_blobs(hasblob, init_blobs, reduce_blob):
    if hasblob or init_blobs is not None or reduce_blob is not None:
        raise NotImplementedError("blobs are arbitrary host objects and cannot cross the device boundary "
                                  "(hasblob=True is not supported by the HIP samplers)")


def metropolis(pdf, sample_ppdf, theta0, niter: int = 10 ** 5, nburnin=None, nthin: int = 1,
               use_progress_meter: bool = True, hasblob: bool = False, init_blobs=None, reduce_blob=None,
               seed=None, device: int = 0):
    """One Metropolis chain with the reference's signature (``src/samplers.jl:59-77``).

    Returns ``(thetas, accept_ratio, logdensities, blobs)`` (``:128``): ``thetas[k]`` is stored sample ``k``
    (shape ``[nsamples]`` for a scalar ``theta0``, ``[nsamples, ndim]`` for a vector), ``accept_ratio`` a float,
    ``blobs = None``.  ``use_progress_meter`` is accepted and ignored (the chain runs in one device call).
    """
    _no_blobs(hasblob, init_blobs, reduce_blob)
    if nburnin is None:
        nburnin = niter // 2                                              # :63
    scalar = np.ndim(theta0) == 0
    th = np.array(theta0, dtype=np.float64).reshape(1, -1)                # :68 deepcopy
    if seed is None:
        seed = _fresh_seed()
    r = run_chains(pdf, sample_ppdf, th, niter, nburnin, nthin, seed, device)
    thetas = r["chain"][:, 0, 0] if scalar else r["chain"][:, 0, :]
    return np.ascontiguousarray(thetas), float(r["accept_ratio"][0]), np.ascontiguousarray(r["chain_logp"][:, 0]), None


def metropolis_chains(pdf, sample_ppdf, theta0s, niter: int = 10 ** 5, nburnin=None, nthin: int = 1,
                      hasblob: bool = False, seed=None, device: int = 0):
    """Many independent Metropolis chains, one per row of ``theta0s`` (``[nchains]`` scalars or
    ``[nchains, ndim]``); ``niter``/``nburnin`` count steps PER CHAIN, as in ``metropolis``.

    Returns ``(thetas, accept_ratio, logdensities, None)`` shaped like ``emcee``'s output
    (``thetas[chain][sample]``), so ``squash_walkers(*result)`` concatenates the chains.
    """
    _no_blobs(hasblob, None, None)
    if nburnin is None:
        nburnin = niter // 2
    theta0s = np.array(theta0s, dtype=np.float64)
    scalar = theta0s.ndim == 1
    if scalar:
        theta0s = theta0s[:, None]
    if seed is None:
        seed = _fresh_seed()
    r = run_chains(pdf, sample_ppdf, theta0s, niter, nburnin, nthin, seed, device)
    thetas = np.ascontiguousarray(r["chain"].transpose(1, 0, 2))
    if scalar:
        thetas = thetas[:, :, 0]
    return thetas, r["accept_ratio"], np.ascontiguousarray(r["chain_logp"].T), None
