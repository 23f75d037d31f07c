"""GPU leg: the HIP path, through the C ABI, against the committed golden fixtures."""
import ctypes as C

import numpy as np
import pytest

import goldenlib

pytestmark = pytest.mark.gpu

_DENS = {0: lambda k, p: k.GaussianIso(p[0], p[1]), 1: lambda k, p: k.Exponential(p[0]),
         2: lambda k, p: k.Rosenbrock(p[0], p[1], p[2])}


@pytest.mark.parametrize("name", goldenlib.names())
def test_sampler_reproduces_golden(kmc, name):
    z = goldenlib.load(name)
    pdf = _DENS[z["density"]](kmc, z["params"])
    with kmc.Sampler(pdf, z["nwalkers"], z["ndim"], z["G"], z["nburnin"], z["nthin"], z["a_scale"], z["seed"],
                     store_chain=True, store_logp=True, moments=True, dtype="f32" if z["f32"] else "f64") as s:
        s.set_positions(z["theta0"])
        s.run(z["G"])
        s.sync()
        chain, chain_logp = s.chain()
        msum, msq, n = s.moments()
        goldenlib.compare(z, s.positions(), s.logp(), s.naccept(), msum, msq, n, chain, chain_logp)


@pytest.mark.parametrize("by_walker", [False, True], ids=["sample-major", "by-walker"])
@pytest.mark.parametrize("name", ["gauss_64x4", "expo_100x1_readme", "rosen_256x64", "rosen_256x64_f32"])
def test_one_shot_c_abi_reproduces_golden(kmc, name, by_walker):
    """kmc_emcee_run with caller-owned host buffers (the entry point a ccall binding uses); with KMC_CHAIN_BY_WALKER the
    chain arrives in the reference's order, thetas[w][k]."""
    from kissmcmc_jl_amd import _lib
    z = goldenlib.load(name)
    nw, nd = z["nwalkers"], z["ndim"]
    ns = (z["G"] - z["nburnin"]) // z["nthin"]
    cfg = _lib.Config()
    cfg.dtype, cfg.density = (_lib.F32 if z["f32"] else _lib.F64), z["density"]
    for i, v in enumerate(z["params"]):
        cfg.params[i] = float(v)
    cfg.nwalkers, cfg.ndim, cfg.ngenerations, cfg.nburnin, cfg.nthin = nw, nd, z["G"], z["nburnin"], z["nthin"]
    cfg.a_scale, cfg.seed, cfg.flags, cfg.device = z["a_scale"], z["seed"], (_lib.CHAIN_BY_WALKER if by_walker else 0), 0
    dp = C.POINTER(C.c_double)
    chain = np.zeros((nw, ns, nd) if by_walker else (ns, nw, nd)); clogp = np.zeros((nw, ns) if by_walker else (ns, nw)); acc = np.zeros(nw)
    nacc = np.zeros(nw, dtype=np.int64); fpos = np.zeros((nw, nd)); flogp = np.zeros(nw)
    msum = np.zeros(nd); msq = np.zeros(nd)
    out = _lib.Outputs()
    out.chain, out.chain_logp = chain.ctypes.data_as(dp), clogp.ctypes.data_as(dp)
    out.accept_ratio, out.naccept = acc.ctypes.data_as(dp), nacc.ctypes.data_as(C.POINTER(C.c_int64))
    out.final_pos, out.final_logp = fpos.ctypes.data_as(dp), flogp.ctypes.data_as(dp)
    out.sum, out.sumsq = msum.ctypes.data_as(dp), msq.ctypes.data_as(dp)
    th = np.ascontiguousarray(z["theta0"], dtype=np.float64)
    th_before = th.copy()
    _lib.check(_lib.lib().kmc_emcee_run(C.byref(cfg), th.ctypes.data_as(dp), C.byref(out)))
    assert (th == th_before).all()                      # the caller's array is never mutated (samplers.jl:198)
    assert out.nsamples == ns and out.device_ms >= 0.0
    if by_walker:
        chain, clogp = chain.transpose(1, 0, 2), clogp.T
    goldenlib.compare(z, fpos, flogp, nacc, msum, msq, out.nmoment, chain, clogp)
    np.testing.assert_array_equal(acc, nacc / (z["G"] - z["nburnin"]))   # samplers.jl:291
