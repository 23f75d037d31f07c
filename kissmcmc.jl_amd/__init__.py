"""kissmcmc.jl_amd -- MI355X-native emcee (affine-invariant ensemble sampler) hot path.

A from-scratch HIP/gfx950 implementation of the one data-parallel path of mauro3/KissMCMC.jl:
``emcee`` / ``_emcee`` (reference ``src/samplers.jl:188-293``), behind the reference's own call
surface (``emcee``, ``make_theta0s``, ``squash_walkers``).  Import as ``kissmcmc_jl_amd``.

The compute path is ``libkissmcmc_hip.so`` (C ABI: ``include/kissmcmc_hip.h``); there is no CPU
fallback -- without the library or a HIP device the sampler raises.
"""
from . import _lib
from ._lib import KmcError
from .api import emcee, emcee_counts, make_theta0s, squash_walkers
from .densities import (CDensity, DeviceLogPdf, Exponential, ExprDensity, GaussianIso, HostLogPdf, LogNormal, MvNormal2,
                        Rosenbrock)
from .diagnostics import eff_samples, int_acorr
from .metropolis import GaussianStep, HostProposal, metropolis, metropolis_chains
from .sampler import Sampler

__all__ = [
    "emcee", "make_theta0s", "squash_walkers", "emcee_counts", "Sampler", "KmcError",
    "DeviceLogPdf", "GaussianIso", "Exponential", "Rosenbrock", "LogNormal", "MvNormal2", "ExprDensity", "CDensity", "HostLogPdf",
    "cdf_g_inv", "g_pdf", "metropolis", "metropolis_chains", "GaussianStep", "HostProposal", "int_acorr", "eff_samples",
]


def g_pdf(z: float, a: float) -> float:
    """Stretch-factor density, reference ``src/samplers.jl:224``."""
    return _lib.lib().kmc_g_pdf(float(z), float(a))


def cdf_g_inv(u: float, a: float) -> float:
    """Inverse CDF of ``g``, reference ``src/samplers.jl:227``."""
    return _lib.lib().kmc_cdf_g_inv(float(u), float(a))
