"""Host cost of the launch modes (table graph vs updated graph) on C2: enqueue wall time against device time, for
several run lengths.  Usage (GPU box): python scripts/exp/launch_host_cost.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc

nw, nd = 65536, 32
th = np.random.default_rng(0).standard_normal((nw, nd))
for mode in ("graph", "updated", ""):
    if mode:
        os.environ["KMC_LAUNCH"] = mode
    else:
        os.environ.pop("KMC_LAUNCH", None)
    for G in (2000, 10000, 20000):
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, 1, 2.0, 12345, moments=True) as s:
            s.set_positions(th)
            s.run(1000); s.sync()
            s.set_positions(th)
            t0 = time.perf_counter()
            s.run(G)
            t1 = time.perf_counter()
            s.sync()
            t2 = time.perf_counter()
            ms = s.last_run_ms()
            print(f"mode={mode or 'auto':8s} G={G:6d} enqueue {1e3*(t1-t0):8.2f} ms  wall {1e3*(t2-t0):8.2f} ms  device {ms:8.2f} ms  "
                  f"{ms/(2*G)*1e3:.3f} us/launch   [{s.describe()[-90:]}]", flush=True)
