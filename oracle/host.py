"""CPU ORACLE (test infrastructure) -- pure-Python restatements of the reference's host-side
pre/post-processing for the emcee path.  Loops are written the way the reference writes them,
line for line, so they are slow: use on small cases only.

* ``make_theta0s``   -- reference ``src/samplers.jl:311-349``
* ``squash_walkers`` -- reference ``src/samplers.jl:372-428``
* ``emcee_counts``   -- the integer bookkeeping of ``emcee`` (``src/samplers.jl:203-204, :234``)

The reference draws from Julia's unseeded default RNG; here the normal stream is a
``numpy.random.Generator`` supplied by the caller, drawn in the same order
(one ``randn()`` for a scalar walker, one ``randn(npara)`` for a vector walker, per try).
"""
from __future__ import annotations

import math
import statistics

import numpy as np


def emcee_counts(niter, nwalkers, nburnin=None, nthin=1):
    """src/samplers.jl:190 (nburnin=niter÷2), :203-204, :234."""
    if nburnin is None:
        nburnin = niter // 2
    niter_walker = niter // nwalkers
    nburnin_walker = nburnin // nwalkers
    nsamples_walker = (niter_walker - nburnin_walker) // nthin
    return niter_walker, nburnin_walker, nsamples_walker


def make_theta0s(theta0, ball_radius, pdf, nwalkers, rng, ball_radius_halfing_steps=7, ntries=100):
    """src/samplers.jl:311-349, including its quirks: ``ball_radius`` is never
    reset (:326), and when every try fails nothing is pushed and no error is raised (:343-345:
    the ``j==ntries`` guard reads the outer ``j = 0``)."""
    scalar = np.ndim(theta0) == 0
    npara = 1 if scalar else len(theta0)                       # :315
    if np.ndim(ball_radius) == 0 and not scalar:               # :316-318
        ball_radius = np.ones(npara) * ball_radius
    assert np.size(ball_radius) == npara                       # :319
    if np.ndim(ball_radius) != 0:
        ball_radius = np.asarray(ball_radius, dtype=np.float64)
    if not scalar:
        theta0 = np.asarray(theta0, dtype=np.float64)
    theta0s = []                                               # :321
    for i in range(1, nwalkers + 1):                           # :323
        for k in range(1, ball_radius_halfing_steps + 1):      # :324
            j = 0                                              # :325
            ball_radius = ball_radius * (1 / 2 ** (k - 1))     # :326
            for _j in range(1, ntries + 1):                    # :327 (inner j shadows the outer one)
                if npara == 1:                                 # :328-332
                    tmp = theta0 + rng.standard_normal() * ball_radius
                else:
                    tmp = theta0 + rng.standard_normal(npara) * ball_radius
                p0 = pdf(tmp)                                  # :336
                if p0 > -math.inf:                             # :338
                    theta0s.append(tmp)                        # :339
                    break
            if len(theta0s) == i:                              # :343
                break
            if j == ntries and k == ball_radius_halfing_steps:  # :344 (never true)
                raise RuntimeError("Could not find suitable initial theta.  PDF is zero in too many places inside ball.")
    return theta0s


def squash_walkers(thetas, accept_ratio, logdensities=None, blobs=None, drop_low_accept_ratio=False,
                   drop_fact=2, order=False):
    """src/samplers.jl:372-428 on list-of-lists inputs (``thetas[w][k]``)."""
    nwalkers = len(accept_ratio)                               # :379
    if drop_low_accept_ratio:                                  # :380-393
        walkers2keep = []
        ma = statistics.median(accept_ratio)
        sa = statistics.stdev(accept_ratio)                    # Julia std: n-1 normalisation
        for nc in range(nwalkers):
            if accept_ratio[nc] <= ma - drop_fact * sa:
                continue
            walkers2keep.append(nc)
    else:
        walkers2keep = list(range(nwalkers))                   # :394-396
    t = list(thetas[walkers2keep[0]])                          # :398
    for w in walkers2keep[1:]:                                 # :399
        t.extend(thetas[w])
    if logdensities is None:                                   # :401-406
        l = None
    else:
        l = list(logdensities[walkers2keep[0]])
        for w in walkers2keep[1:]:
            l.extend(logdensities[w])
    b = None                                                   # blobs: out of scope
    if order:                                                  # :415-426
        nc = len(walkers2keep)
        ns = len(thetas[0])
        keys = []
        for _ in range(nc):
            keys.extend(range(1, ns + 1))
        perm = sorted(range(len(keys)), key=lambda i: keys[i])  # sortperm: stable
        if l is not None:
            l = [l[i] for i in perm]
        t = [t[i] for i in perm]
    mean_acc = sum(accept_ratio[w] for w in walkers2keep) / len(walkers2keep)   # :427
    return t, mean_acc, l, b


# ---- integrated autocorrelation time: the (commented-out) int_acorr / acor1d / auto_window / eff_samples of
#      reference src/analysis.jl:140-167 (int_acorr), :185-191 (eff_samples), :252-273 (acor1d), :280-285 (auto_window).
#      The reference file is 100 % commented out -- there is no live behaviour and no reference test for it; this
#      restates the code as written (circular autocorrelation: the FFT is NOT zero-padded, :258-260), numpy FFT. ----
def acor1d(x, norm=True):
    """reference src/analysis.jl:252-273"""
    import numpy as np
    x = np.asarray(x, dtype=np.float64)
    f = np.fft.fft(x - x.mean())                       # :258
    acf = np.real(np.fft.ifft(f * np.conj(f)))         # :259
    acf = acf / (4 * len(x))                           # :260
    if norm:
        acf = acf / acf[0]                             # :264
    return acf[: len(acf) // 2]                        # :267


def auto_window(taus, c):
    """reference src/analysis.jl:280-285 (1-based i, returned 0-based here)"""
    for i, t in enumerate(taus, start=1):
        if i >= c * t:
            return i - 1
    return len(taus) - 2                               # length(taus)-1, 1-based


def int_acorr(thetas, c=5):
    """reference src/analysis.jl:140-167; thetas[ntheta][nsamples][nchains].  Returns (tau[ntheta], converged[ntheta])."""
    import numpy as np
    thetas = np.asarray(thetas, dtype=np.float64)
    assert c > 1                                       # :141
    ntheta, nsamples, nchains = thetas.shape           # :143
    out = []
    for n in range(ntheta):                            # :145
        rho = np.zeros(nsamples // 2)                  # :147
        for cc in range(nchains):
            rho += acor1d(thetas[n, :, cc])            # :149
        rho /= nchains                                 # :151
        taus = 2 * np.cumsum(rho) - 1                  # :153
        out.append(taus[auto_window(taus, c)])         # :154-155
    out = np.array(out)
    converged = nsamples / out                         # :157
    if np.any(np.isnan(out)) or np.any(np.isnan(converged)):   # :161-165
        out = out * 0 - 1
        converged = converged * 0 - 1
    return out, converged


def eff_samples(thetas, c=5):
    """reference src/analysis.jl:185-191"""
    import numpy as np
    thetas = np.asarray(thetas, dtype=np.float64)
    acorr, converged = int_acorr(thetas, c=c)
    ns = thetas.shape[1] / acorr * thetas.shape[2]
    return (int(round(float(np.mean(ns)))), int(round((thetas.shape[1] * thetas.shape[2]) // float(np.mean(ns)))),
            float(np.mean(converged)), np.round(ns).astype(np.int64), acorr, converged)
