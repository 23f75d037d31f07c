import sys, os, time
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc
import torch
def rss():
    for l in open('/proc/self/status'):
        if l.startswith('VmRSS'): return int(l.split()[1]) / 1024
pdf = kmc.Exponential(1.0)
th0 = kmc.make_theta0s(0.5, 0.1, pdf, 100, rng=3)
cd = kmc.CDensity("return x[0] >= 0 ? -x[0] : -INFINITY;")
for name, p in (("menu", pdf), ("cdensity", cd)):
    kmc.emcee(p, th0, niter=10 ** 5, seed=1, use_progress_meter=False)
    r0, f0 = rss(), torch.cuda.mem_get_info(0)[0]
    t0 = time.time()
    N = 3000
    for i in range(N):
        kmc.emcee(p, th0, niter=10 ** 5, seed=i, use_progress_meter=False)
    print(f"{name}: {N} README calls in {time.time() - t0:.1f} s; host RSS {r0:.0f} -> {rss():.0f} MiB; device free changed by {(f0 - torch.cuda.mem_get_info(0)[0]) / 2**20:.1f} MiB", flush=True)
from kissmcmc_jl_amd.metropolis import GaussianStep
r0 = rss(); t0 = time.time()
for i in range(500):
    kmc.metropolis(kmc.GaussianIso(-5.0, 3.0), GaussianStep(5.0), 0.0, niter=10 ** 4, seed=i, use_progress_meter=False)
print(f"metropolis: 500 calls in {time.time() - t0:.1f} s; host RSS {r0:.0f} -> {rss():.0f} MiB")
