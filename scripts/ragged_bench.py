"""Half-step time for arbitrary ndim: masked vector kernel vs the generic one-walker-per-lane kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc

nw = 65536
for nd in (1, 2, 3, 10, 20, 50, 100, 200, 500):
    th = np.random.default_rng(0).standard_normal((nw, nd))
    for plan in ("", "generic"):
        if plan:
            os.environ["KMC_PLAN"] = plan
        else:
            os.environ.pop("KMC_PLAN", None)
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, 10 ** 9, 0, 1, 2.0, 1, moments=True) as s:
            s.set_positions(th)
            s.run(64)
            s.sync()
            ts = []
            for _ in range(3):
                s.run(128)
                s.sync()
                ts.append(s.last_run_ms())
            t = min(ts)
            name = plan or "vec"
            print(f"ndim={nd:4d} plan={name:8s} {t / 256 * 1e3:8.2f} us/half-step  {nw * 128 / t / 1e6:7.3f} Gsteps/s", flush=True)
