"""Is the updated-graph mode fed fast enough by the host?  Host wall time of kmc_sampler_run (enqueue only) against the HIP-event
time of the same generations, C2 shape.  python scripts/exp/host_feed_check.py [generations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
th = np.random.default_rng(0).standard_normal((65536, 32))
os.environ["KMC_DEBUG"] = "feed-stats"
PDFS = {"menu": lambda: kmc.GaussianIso(),
        "sum-body": lambda: kmc.CDensity("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"),
        "coupled-body": lambda: kmc.CDensity("double s = 0, t = 0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += x[i]; } return -0.5 * (s + p[0] * t * t);", params=[0.05])}
for name, make in PDFS.items():
  for mode in (None,):
    pdf = make()
    with kmc.Sampler(pdf, 65536, 32, 10 ** 9, 0, 1, 2.0, 7, moments=True) as s:
        s.set_positions(th)
        s.run(2048)
        s.sync()
        for rep in range(2):
            t0 = time.perf_counter()
            s.run(G)
            t1 = time.perf_counter()
            s.sync()
            t2 = time.perf_counter()
            print(f"{name}: {G} generations: enqueue {1e3 * (t1 - t0):8.2f} ms, until done {1e3 * (t2 - t0):8.2f} ms, HIP events {s.last_run_ms():8.2f} ms "
                  f"= {s.last_run_ms() / (2 * G) * 1e3:.3f} us per half-step; {s.describe()[-260:]}", flush=True)
