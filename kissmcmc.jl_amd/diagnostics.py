"""Convergence diagnostics on the samplers' output: the integrated autocorrelation time ``int_acorr`` and
``eff_samples`` of reference ``src/analysis.jl:140-167, :185-191`` (with ``acor1d`` ``:252-273`` and ``auto_window``
``:280-285``).  In the reference that file is entirely commented out -- the README's "check convergence using integrated
autocorrelation" (``README.md:26``) has no live code behind it -- so the code is followed as written and there is no
reference behaviour to be bit-compatible with.  The FFTs and reductions run on the GPU (``kmc_int_acorr``).

Input layout: what ``emcee`` / ``metropolis_chains`` return, ``thetas[walker][sample]`` (scalar walkers) or
``thetas[walker][sample][dim]`` -- the reference's ``(ntheta, nsamples, nchains)`` array with the axes reversed.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib


def int_acorr(thetas, c: float = 5, warn: bool = True, warnat: float = 50, device: int = 0):
    """Integrated autocorrelation time tau (in stored samples) per dimension and ``nsamples / tau`` ("converged",
    reference ``:157``: should probably be larger than 50 or 100).  Returns ``(tau[ndim], converged[ndim])``."""
    if not c > 1:
        raise AssertionError("c>1")                                           # :141
    th = np.asarray(thetas, dtype=np.float64)
    if th.ndim == 2:
        th = th[:, :, None]
    if th.ndim != 3:
        raise ValueError("thetas must be [walker][sample] or [walker][sample][dim]")
    nwalkers, nsamples, ndim = th.shape
    chain = np.ascontiguousarray(th.transpose(1, 0, 2))                       # [sample][walker][dim], the device chain layout
    tau = np.zeros(ndim)
    conv = np.zeros(ndim)
    dp = C.POINTER(C.c_double)
    _lib.check(_lib.lib().kmc_int_acorr(chain.ctypes.data_as(dp), nsamples, nwalkers, ndim, float(c), int(device),
                                        tau.ctypes.data_as(dp), conv.ctypes.data_as(dp)))
    if warn and np.any(conv < warnat):                                        # :158-160
        warnings.warn("Estimate of integrated autocorrelation likely not accurate!")
    return tau, conv


def eff_samples(thetas, c: float = 5, device: int = 0):
    """reference ``src/analysis.jl:185-191``: ``(Neff, suggested thinning, mean convergence estimate, Neff per dimension,
    tau per dimension, convergence estimate per dimension)``."""
    th = np.asarray(thetas)
    nwalkers, nsamples = th.shape[0], th.shape[1]
    acorr, converged = int_acorr(thetas, c=c, warn=False, device=device)
    ns = nsamples / acorr * nwalkers                                          # :187
    return (int(round(float(np.mean(ns)))), int(round((nsamples * nwalkers) // float(np.mean(ns)))), float(np.mean(converged)),
            np.round(ns).astype(np.int64), acorr, converged)
