"""Runtime-compiled kernels run 7-19 % behind the compiled-in menu kernel of identical ISA: launch path or code object?
menu (compiled in) / menu launched through hipFunction_t + argument buffer (KMC_DEBUG=menu-via-module) / the same density as an
ExprDensity (hiprtc), each under the table graph (KMC_LAUNCH=graph) and eager.  python scripts/exp/module_path_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc

G = 4096
th = np.random.default_rng(0).standard_normal((65536, 32))
for launch in ("graph", "eager", "updated"):
    os.environ["KMC_LAUNCH"] = launch
    for name, dbg, make in (("menu", "", lambda: kmc.GaussianIso()), ("menu via module launch", "menu-via-module", lambda: kmc.GaussianIso()),
                            ("ExprDensity (hiprtc)", "rtc=hiprtc", lambda: kmc.ExprDensity("-0.5*((x-p[0])*p[1])*((x-p[0])*p[1])", None, [0.0, 1.0])),
                            ("ExprDensity (hipcc child)", "rtc=hipcc", lambda: kmc.ExprDensity("-0.5*((x-p[0])*p[1])*((x-p[0])*p[1])", None, [0.0, 1.0]))):
        os.environ["KMC_DEBUG"] = dbg
        if launch == "updated" and dbg == "menu-via-module":
            continue
        with kmc.Sampler(make(), 65536, 32, 10 ** 9, 0, 1, 2.0, 7, moments=True) as s:
            s.set_positions(th)
            s.run(1024)
            s.sync()
            ts = []
            for rep in range(3):
                s.run(G)
                s.sync()
                ts.append(s.last_run_ms() / (2 * G) * 1e3)
            print(f"KMC_LAUNCH={launch:8s} {name:26s}: {min(ts):.3f} us per half-step (runs {', '.join(f'{t:.3f}' for t in ts)})", flush=True)
