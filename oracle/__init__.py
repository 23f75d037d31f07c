"""CPU ORACLE -- test infrastructure, NOT product code.

ctypes bindings for ``oracle/kmc_oracle.c`` (a plain-C fp64 restatement of the reference's
``emcee`` hot path, reference ``src/samplers.jl:188-293``) plus pure-Python restatements of the
host-side pre/post-processing (``make_theta0s`` ``src/samplers.jl:311-349``, ``squash_walkers``
``src/samplers.jl:372-428``) in :mod:`oracle.host`.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Parity status: pinned against the reference's own known-answer and statistical
tests (``tests/test_oracle_pins.py``); bit-level parity with the reference is undefined because
the reference never seeds its RNG (see the header of ``kmc_oracle.c``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libkmc_oracle.so")

GAUSSIAN_ISO, EXPONENTIAL, ROSENBROCK, LOGNORMAL, MVNORMAL2 = 0, 1, 2, 3, 4
OK, ERR_A_SCALE, ERR_ODD_WALKERS, ERR_TOO_FEW_WALKERS, ERR_BAD_ARG, ERR_NONFINITE_LOGP = range(6)


class Config(C.Structure):
    _fields_ = [
        ("density", C.c_int32),
        ("nthreads", C.c_int32),
        ("params", C.c_double * 8),
        ("nwalkers", C.c_int64),
        ("ndim", C.c_int64),
        ("ngenerations", C.c_int64),
        ("nburnin", C.c_int64),
        ("nthin", C.c_int64),
        ("a_scale", C.c_double),
        ("seed", C.c_uint64),
        ("state_f32", C.c_int32),
        ("pad_", C.c_int32),
    ]


class MetropolisConfig(C.Structure):
    _fields_ = [
        ("density", C.c_int32),
        ("nthreads", C.c_int32),
        ("params", C.c_double * 8),
        ("nchains", C.c_int64),
        ("ndim", C.c_int64),
        ("niter", C.c_int64),
        ("nburnin", C.c_int64),
        ("nthin", C.c_int64),
        ("seed", C.c_uint64),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "kmc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        L.kmco_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.kmco_g_pdf.restype = C.c_double
        L.kmco_g_pdf.argtypes = [C.c_double, C.c_double]
        L.kmco_cdf_g_inv.restype = C.c_double
        L.kmco_cdf_g_inv.argtypes = [C.c_double, C.c_double]
        L.kmco_sample_g.restype = C.c_double
        L.kmco_sample_g.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_double]
        L.kmco_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64,
                                C.POINTER(C.c_int64), dp, dp]
        L.kmco_accept_terms.restype = None
        L.kmco_accept_terms.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                        C.POINTER(C.c_int64), dp, dp, dp]
        L.kmco_logpdf.restype = C.c_double
        L.kmco_logpdf.argtypes = [C.c_int32, dp, dp, C.c_int64]
        L.kmco_validate.argtypes = [C.POINTER(Config)]
        L.kmco_half_step.restype = None
        L.kmco_half_step.argtypes = [C.POINTER(Config), dp, dp, C.POINTER(C.c_int64),
                                     C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int]
        L.kmco_emcee.argtypes = [C.POINTER(Config), dp, dp, dp, dp, C.POINTER(C.c_int64),
                                 dp, dp, dp, dp, C.POINTER(C.c_int64)]
        L.kmco_island_perm.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.kmco_emcee_islands.argtypes = [C.POINTER(Config), C.c_int64, C.c_int64, dp, dp, C.POINTER(C.c_int64),
                                         dp, dp, dp, dp, C.POINTER(C.c_int64)]
        assert L.kmco_sizeof_config() == C.sizeof(Config)
        L.kmco_deal_seed.restype = C.c_uint64
        L.kmco_deal_seed.argtypes = [C.c_uint64, C.c_int32]
        L.kmco_deal_perm.restype = None
        L.kmco_deal_perm.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.kmco_emcee_dealt_chain.argtypes = [C.POINTER(Config), C.c_int32, C.c_int64, dp, dp, C.POINTER(C.c_int64), dp, dp,
                                             C.POINTER(C.c_int64), dp, dp, C.POINTER(C.c_int64), dp, dp]
        L.kmco_emcee_dealt.argtypes = [C.POINTER(Config), C.c_int32, C.c_int64, dp, dp, C.POINTER(C.c_int64), dp, dp,
                                       C.POINTER(C.c_int64), dp, dp, C.POINTER(C.c_int64)]
        L.kmco_init_ball.restype = C.c_int64
        L.kmco_init_ball.argtypes = [C.c_int32, dp, dp, dp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint64,
                                     dp, dp, C.POINTER(C.c_int64)]
        L.kmco_metropolis_draw.restype = None
        L.kmco_metropolis_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, dp, dp]
        L.kmco_metropolis.argtypes = [C.POINTER(MetropolisConfig), dp, dp, dp, dp, dp, C.POINTER(C.c_int64), dp, dp, dp, dp]
        assert L.kmco_sizeof_metropolis_config() == C.sizeof(MetropolisConfig)
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int64))


def make_config(density, params, nwalkers, ndim, ngenerations=0, nburnin=0, nthin=1,
                a_scale=2.0, seed=0, nthreads=1, state_f32=False) -> Config:
    c = Config()
    c.density = int(density)
    c.nthreads = int(nthreads)
    p = list(params) + [0.0] * (8 - len(params))
    for i in range(8):
        c.params[i] = float(p[i])
    c.nwalkers, c.ndim = int(nwalkers), int(ndim)
    c.ngenerations, c.nburnin, c.nthin = int(ngenerations), int(nburnin), int(nthin)
    c.a_scale, c.seed = float(a_scale), int(seed)
    c.state_f32 = 1 if state_f32 else 0
    return c


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().kmco_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def g_pdf(z, a):
    return lib().kmco_g_pdf(z, a)


def cdf_g_inv(u, a):
    return lib().kmco_cdf_g_inv(u, a)


def sample_g(seed, step, walker, a):
    return lib().kmco_sample_g(seed, step, walker, a)


def draw(seed, step, walker, nhalf):
    p = C.c_int64()
    uz = C.c_double()
    ua = C.c_double()
    lib().kmco_draw(seed, step, walker, nhalf, C.byref(p), C.byref(uz), C.byref(ua))
    return p.value, uz.value, ua.value


def accept_terms(seed, step, walker0, n, nhalf, a_scale, ndim):
    """``(partner, z, t1, lu)`` of walkers ``walker0 .. walker0 + n - 1`` at ``step``: the random side of the accept test
    (``src/samplers.jl:260``) with the oracle's arithmetic (glibc ``log``)."""
    part = np.empty(n, dtype=np.int64)
    z, t1, lu = np.empty(n), np.empty(n), np.empty(n)
    lib().kmco_accept_terms(seed, step, walker0, n, nhalf, float(a_scale), float(ndim - 1), _ip(part), _dp(z), _dp(t1), _dp(lu))
    return part, z, t1, lu


def logpdf(density, params, x):
    x = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
    p = np.zeros(8)
    p[: len(params)] = params
    return lib().kmco_logpdf(int(density), _dp(p), _dp(x), x.size)


def logpdf_batch(density, params, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    return np.array([logpdf(density, params, row) for row in X.reshape(X.shape[0], -1)])


def half_step(cfg: Config, pos, logp, naccept, generation, half, active_begin=0, n_active=None,
              count_accept=True):
    """In-place half-step on ``pos [nw,nd]``, ``logp [nw]``, ``naccept [nw] int64``."""
    assert pos.dtype == np.float64 and pos.flags.c_contiguous
    assert logp.dtype == np.float64 and naccept.dtype == np.int64
    if n_active is None:
        n_active = cfg.nwalkers // 2
    lib().kmco_half_step(C.byref(cfg), _dp(pos), _dp(logp), _ip(naccept), int(generation), int(half),
                         int(active_begin), int(n_active), int(bool(count_accept)))


def emcee(cfg: Config, theta0, store_chain=True, moments=True):
    """Run the oracle sampler.  Returns a dict of dense arrays (see kmc_oracle.c: kmco_emcee)."""
    nw, nd = cfg.nwalkers, cfg.ndim
    theta0 = np.ascontiguousarray(np.asarray(theta0, dtype=np.float64).reshape(nw, nd))
    ns = max(0, (cfg.ngenerations - cfg.nburnin) // cfg.nthin)
    chain = np.zeros((ns, nw, nd)) if store_chain else None
    chain_logp = np.zeros((ns, nw)) if store_chain else None
    acc = np.zeros(nw)
    nacc = np.zeros(nw, dtype=np.int64)
    fpos = np.zeros((nw, nd))
    flogp = np.zeros(nw)
    msum = np.zeros(nd) if moments else None
    msq = np.zeros(nd) if moments else None
    nmom = C.c_int64(0)
    with np.errstate(all="ignore"):
        st = lib().kmco_emcee(C.byref(cfg), _dp(theta0), _dp(chain), _dp(chain_logp), _dp(acc), _ip(nacc),
                              _dp(fpos), _dp(flogp), _dp(msum), _dp(msq), C.byref(nmom))
    return dict(status=st, chain=chain, chain_logp=chain_logp, accept_ratio=acc, naccept=nacc,
                final_pos=fpos, final_logp=flogp, sum=msum, sumsq=msq, nmoment=nmom.value,
                nsamples=ns)


def init_ball(density, params, theta0, ball_radius, nrows, ndim, seed=0, halving_steps=7, ntries=100, walker0=0):
    """Seeded initial ball (kmc_oracle.c: kmco_init_ball; reference src/samplers.jl:311-349, per-walker shrink).
    Returns dict(pos [nrows, ndim], logp [nrows], attempts [nrows] (tries used, -1 = none admissible), nfail)."""
    p = np.zeros(8)
    p[: len(params)] = params
    th = np.ascontiguousarray(np.broadcast_to(np.asarray(theta0, dtype=np.float64), (ndim,)))
    rad = np.ascontiguousarray(np.broadcast_to(np.asarray(ball_radius, dtype=np.float64), (ndim,)))
    pos = np.zeros((nrows, ndim))
    logp = np.zeros(nrows)
    att = np.zeros(nrows, dtype=np.int64)
    with np.errstate(all="ignore"):
        nfail = lib().kmco_init_ball(int(density), _dp(p), _dp(th), _dp(rad), int(nrows), int(walker0), int(ndim),
                                     int(halving_steps), int(ntries), int(seed) & 0xFFFFFFFFFFFFFFFF, _dp(pos), _dp(logp), _ip(att))
    return dict(pos=pos, logp=logp, attempts=att, nfail=int(nfail))


def island_perm(seed, epoch, nwalkers):
    a = C.c_int64()
    c = C.c_int64()
    lib().kmco_island_perm(seed, epoch, nwalkers, C.byref(a), C.byref(c))
    return a.value, c.value


def emcee_islands(cfg: Config, island_size, epoch_gens, theta0, moments=True):
    """Island-mode oracle (see kmc_oracle.c: kmco_emcee_islands)."""
    nw, nd = cfg.nwalkers, cfg.ndim
    theta0 = np.ascontiguousarray(np.asarray(theta0, dtype=np.float64).reshape(nw, nd))
    acc = np.zeros(nw)
    nacc = np.zeros(nw, dtype=np.int64)
    fpos = np.zeros((nw, nd))
    flogp = np.zeros(nw)
    msum = np.zeros(nd) if moments else None
    msq = np.zeros(nd) if moments else None
    nmom = C.c_int64(0)
    with np.errstate(all="ignore"):
        st = lib().kmco_emcee_islands(C.byref(cfg), int(island_size), int(epoch_gens), _dp(theta0), _dp(acc), _ip(nacc),
                                      _dp(fpos), _dp(flogp), _dp(msum), _dp(msq), C.byref(nmom))
    return dict(status=st, accept_ratio=acc, naccept=nacc, final_pos=fpos, final_logp=flogp, sum=msum, sumsq=msq,
                nmoment=nmom.value)


def deal_seed(seed, rank):
    return int(lib().kmco_deal_seed(int(seed) & 0xFFFFFFFFFFFFFFFF, int(rank)))


def deal_perm(seed, epoch, rank, S):
    a = C.c_int64()
    c = C.c_int64()
    lib().kmco_deal_perm(int(seed) & 0xFFFFFFFFFFFFFFFF, int(epoch), int(rank), int(S), C.byref(a), C.byref(c))
    return a.value, c.value


def emcee_dealt(cfg: Config, nsub, epoch_gens, theta0, moments=True, store_chain=False):
    """Dealt sub-ensembles (kmc_oracle.c: kmco_emcee_dealt_chain): ``cfg.nwalkers`` walkers in ``nsub`` sub-ensembles, re-dealt
    every ``epoch_gens`` generations.  Per-walker outputs are in GLOBAL WALKER order; ``slot_ids`` = walker per final slot;
    ``store_chain``: ``chain [nsamples, nwalkers, ndim]`` and ``chain_logp [nsamples, nwalkers]``, by walker as well."""
    nw, nd = cfg.nwalkers, cfg.ndim
    theta0 = np.ascontiguousarray(np.asarray(theta0, dtype=np.float64).reshape(nw, nd))
    acc = np.zeros(nw)
    nacc = np.zeros(nw, dtype=np.int64)
    fpos = np.zeros((nw, nd))
    flogp = np.zeros(nw)
    ids = np.zeros(nw, dtype=np.int64)
    msum = np.zeros(nd) if moments else None
    msq = np.zeros(nd) if moments else None
    nmom = C.c_int64(0)
    ns = (cfg.ngenerations - cfg.nburnin) // cfg.nthin if cfg.ngenerations > cfg.nburnin else 0
    chain = np.zeros((ns, nw, nd)) if store_chain else None
    chain_logp = np.zeros((ns, nw)) if store_chain else None
    with np.errstate(all="ignore"):
        st = lib().kmco_emcee_dealt_chain(C.byref(cfg), int(nsub), int(epoch_gens), _dp(theta0), _dp(acc), _ip(nacc), _dp(fpos), _dp(flogp),
                                          _ip(ids), _dp(msum), _dp(msq), C.byref(nmom), _dp(chain), _dp(chain_logp))
    return dict(status=st, accept_ratio=acc, naccept=nacc, final_pos=fpos, final_logp=flogp, slot_ids=ids, sum=msum, sumsq=msq,
                nmoment=nmom.value, chain=chain, chain_logp=chain_logp)


def metropolis_draw(seed, it, chain, ndim):
    """The ndim standard normals and the accept uniform of (iteration, chain)."""
    nrm = np.zeros(ndim)
    ua = C.c_double()
    lib().kmco_metropolis_draw(int(seed), int(it), int(chain), int(ndim), _dp(nrm), C.byref(ua))
    return nrm, ua.value


def metropolis(density, params, theta0, step, niter, nburnin=None, nthin=1, seed=0, nthreads=1,
               store_chain=True, moments=True):
    """Many-chain Metropolis oracle (kmc_oracle.c: kmco_metropolis; reference src/samplers.jl:59-128 per chain).
    ``theta0 [nchains, ndim]``, ``step`` scalar or ``[ndim]``; ``niter``/``nburnin`` are steps per chain."""
    theta0 = np.ascontiguousarray(np.asarray(theta0, dtype=np.float64))
    nc, nd = theta0.shape
    if nburnin is None:
        nburnin = niter // 2                                       # src/samplers.jl:63
    c = MetropolisConfig()
    c.density, c.nthreads = int(density), int(nthreads)
    p = list(params) + [0.0] * (8 - len(params))
    for i in range(8):
        c.params[i] = float(p[i])
    c.nchains, c.ndim, c.niter, c.nburnin, c.nthin, c.seed = nc, nd, int(niter), int(nburnin), int(nthin), int(seed)
    step = np.ascontiguousarray(np.broadcast_to(np.asarray(step, dtype=np.float64), (nd,)))
    ns = max(0, (niter - nburnin) // nthin)
    chain = np.zeros((ns, nc, nd)) if store_chain else None
    chain_logp = np.zeros((ns, nc)) if store_chain else None
    acc = np.zeros(nc)
    nacc = np.zeros(nc, dtype=np.int64)
    fpos = np.zeros((nc, nd))
    flogp = np.zeros(nc)
    csum = np.zeros((nc, nd)) if moments else None
    csq = np.zeros((nc, nd)) if moments else None
    with np.errstate(all="ignore"):
        st = lib().kmco_metropolis(C.byref(c), _dp(theta0), _dp(step), _dp(chain), _dp(chain_logp), _dp(acc), _ip(nacc),
                                   _dp(fpos), _dp(flogp), _dp(csum), _dp(csq))
    return dict(status=st, chain=chain, chain_logp=chain_logp, accept_ratio=acc, naccept=nacc, final_pos=fpos,
                final_logp=flogp, chain_sum=csum, chain_sumsq=csq, nsamples=ns)
