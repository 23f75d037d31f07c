"""Device time of the Metropolis chains with the draws made in the kernel (KMC_DEBUG=metro-table=0) and read from a table (=1), by chain count."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd.metropolis import run_chains, GaussianStep

for name, pdf, nd in (("gauss 1-D", kmc.GaussianIso(-5.0, 3.0), 1), ("rosen 2-D", kmc.Rosenbrock(1.0, 100.0, 20.0), 2), ("gauss 8-D", kmc.GaussianIso(0.0, 1.0), 8)):
    for nc in (1, 64, 1024, 4096, 16384, 65536):
        niter = 200000 if nc <= 64 else (20000 if nc <= 4096 else 4000)
        row = []
        for mode in ("0", "1"):
            os.environ["KMC_DEBUG"] = "metro-table=" + mode
            r = run_chains(pdf, GaussianStep(1.5), np.zeros((nc, nd)), niter, niter // 2, 10, 3, store_chain=True, store_logp=False, moments=True)
            row.append(r["device_ms"])
        del os.environ["KMC_DEBUG"]
        print(f"{name:10s} {nc:6d} chains x {niter:6d} steps: in-kernel draws {row[0]:9.3f} ms ({nc * niter / (row[0] * 1e-3):.3e} chain-steps/s), "
              f"table {row[1]:9.3f} ms ({nc * niter / (row[1] * 1e-3):.3e}) -> x{row[0] / row[1]:.2f}", flush=True)
