"""Soak of the updated-graph mode with a runtime-compiled vector kernel (module function as graph kernel node, parameters rewritten before
every replay): 200 000 generations of the C2 ensemble with the Gaussian written as a C function body, against the menu density's run of
the same job -- positions, counters and log-pdfs must be equal bit for bit at the end.  python scripts/exp/user_kernel_soak.py [generations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
th = np.random.default_rng(1).standard_normal((65536, 32))
out = {}
for name, pdf in (("menu", kmc.GaussianIso()), ("body", kmc.CDensity("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"))):
    with kmc.Sampler(pdf, 65536, 32, G, G // 2, 1, 2.0, 99, moments=True) as s:
        s.set_positions(th)
        t0 = time.perf_counter()
        s.run(G)
        s.sync()
        out[name] = (s.positions(), s.naccept(), s.logp(), s.moments())
        print(f"{name}: {G} generations in {time.perf_counter() - t0:.2f} s, {s.last_run_ms() / (2 * G) * 1e3:.3f} us per half-step; launch mode {s.launch_mode()}; {s.describe()[-120:]}", flush=True)
same = all(np.array_equal(out["menu"][i], out["body"][i]) for i in range(3)) and out["menu"][3][2] == out["body"][3][2] and np.array_equal(out["menu"][3][0], out["body"][3][0])
print("bit-identical:", same)
sys.exit(0 if same else 1)
