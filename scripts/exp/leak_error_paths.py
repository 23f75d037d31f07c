"""Failing calls (validation errors, a chain beyond HBM, a density that does not compile, a non-finite start) by the thousand:
host RSS and free device memory before and after."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kissmcmc_jl_amd as kmc
def rss_kb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1])
def attempt(i):
    k = i % 5
    try:
        if k == 0: kmc.Sampler(kmc.GaussianIso(), 101, 4, 10)
        elif k == 1: kmc.Sampler(kmc.GaussianIso(), 4, 32, 10)
        elif k == 2: kmc.Sampler(kmc.GaussianIso(), 65536, 32, 200000, 0, 1, store_chain=True)
        elif k == 3: kmc.ExprDensity("-0.5*x*x +")
        else:
            with kmc.Sampler(kmc.Exponential(), 64, 2, 10) as s:
                s.set_positions(-np.ones((64, 2)))          # non-finite initial log-pdf
    except (kmc.KmcError, ValueError, AssertionError):
        pass
for i in range(200): attempt(i)
r0, f0 = rss_kb(), torch.cuda.mem_get_info()[0]
N = 5000
for i in range(N): attempt(i)
print(f"{N} failing calls: RSS {rss_kb() - r0:+d} KiB, free device memory {(torch.cuda.mem_get_info()[0] - f0) / 2**20:+.1f} MiB")
