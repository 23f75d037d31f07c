"""Which kind of sampler leaks host memory per create / run / destroy cycle (see leak_check.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd.metropolis import run_chains
from kissmcmc_jl_amd import _lib
import ctypes as C

def rss_kb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1])
rng = np.random.default_rng(0)
expr = kmc.ExprDensity("-0.5*x*x")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000

def sampler_cycle(pdf, nw, nd, G, run=True, **kw):
    th = rng.standard_normal((nw, nd))
    def f(i):
        with kmc.Sampler(pdf, nw, nd, G, G // 4, 1, 2.0, i + 1, **kw) as s:
            if run:
                s.set_positions(th)
                s.run(G)
                s.sync()
    return f
kinds = [
    ("create + destroy only (multi-launch)", sampler_cycle(kmc.GaussianIso(), 2048, 8, 70, run=False)),
    ("eager run (20 generations)", sampler_cycle(kmc.GaussianIso(), 2048, 8, 20)),
    ("table graph (70 generations)", sampler_cycle(kmc.GaussianIso(), 2048, 8, 70)),
    ("updated graph (900 generations)", sampler_cycle(kmc.GaussianIso(), 2048, 32, 900)),
    ("resident", sampler_cycle(kmc.GaussianIso(), 100, 2, 50)),
    ("runtime-compiled (expr)", sampler_cycle(expr, 2048, 8, 20)),
    ("streamed chain", sampler_cycle(kmc.GaussianIso(), 512, 8, 200, store_chain=True, stream_chain=True)),
    ("metropolis", lambda i: run_chains(kmc.GaussianIso(), kmc.GaussianStep(0.5), rng.standard_normal((256, 3)), 40, 10, 1, i + 1)),
    ("torch stream create only", lambda i: torch.cuda.Stream()),
]
for name, f in kinds:
    for i in range(200):
        f(i)
    torch.cuda.synchronize()
    r0 = rss_kb()
    for i in range(N):
        f(i)
    torch.cuda.synchronize()
    r1 = rss_kb()
    print(f"{name:40s} {N} cycles: RSS {r1 - r0:+8d} KiB  ({(r1 - r0) / N:6.2f} KiB per cycle)", flush=True)
