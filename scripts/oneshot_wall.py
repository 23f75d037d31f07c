import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, ctypes as C
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd import _lib
L = _lib.lib()
nw, nd, G = 65536, 32, 10000
th = np.ascontiguousarray(np.random.default_rng(0).standard_normal((nw, nd)))
for rep in range(3):
    c = _lib.Config()
    c.dtype, c.density = _lib.F64, 0
    c.params[0], c.params[1] = 0.0, 1.0
    c.nwalkers, c.ndim, c.ngenerations, c.nburnin, c.nthin, c.a_scale, c.seed = nw, nd, G, G // 2, 1, 2.0, 5
    out = _lib.Outputs()
    acc = np.zeros(nw); sm = np.zeros(nd); sq = np.zeros(nd); fp = np.zeros((nw, nd))
    dp = C.POINTER(C.c_double)
    out.accept_ratio, out.sum, out.sumsq, out.final_pos = acc.ctypes.data_as(dp), sm.ctypes.data_as(dp), sq.ctypes.data_as(dp), fp.ctypes.data_as(dp)
    t0 = time.perf_counter()
    _lib.check(L.kmc_emcee_run(C.byref(c), th.ctypes.data_as(dp), C.byref(out)))
    t1 = time.perf_counter()
    print(f"kmc_emcee_run C2: wall {1e3 * (t1 - t0):.1f} ms, device loop {out.device_ms:.1f} ms, acc {acc.mean():.4f}", flush=True)
