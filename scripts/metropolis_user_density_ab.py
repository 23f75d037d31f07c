import sys, os
sys.path.insert(0, ".")
import numpy as np
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd.metropolis import run_chains, GaussianStep
body = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"
for nd in (8, 16, 32):
    for nc in (1, 64, 1024):
        row = []
        for mode in ("0", "1"):
            os.environ["KMC_DEBUG"] = "metro-table=" + mode
            r = run_chains(kmc.CDensity(body), GaussianStep(0.5), np.zeros((nc, nd)), 50000 if nc <= 64 else 5000, 100, 10, 3, store_chain=True, store_logp=False, moments=True)
            row.append(r["device_ms"])
        print(f"CDensity {nd}-D, {nc} chains: in-kernel (chain in memory) {row[0]:.2f} ms, table (chain in registers) {row[1]:.2f} ms -> x{row[0] / row[1]:.2f}", flush=True)
