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
def chain_cycle(i):
    with kmc.Sampler(kmc.GaussianIso(), 512, 7, 40, 0, 1, 2.0, i + 1, store_chain=True, store_logp=True) as s:
        s.set_positions(rng.standard_normal((512, 7)))
        s.run(40); s.sync()
        s.chain(); s.chain(by_walker=True)

def dealt_cycle(i):
    from kissmcmc_jl_amd.distributed import HipDealExecutor, LocalDealtEmcee
    exs = [HipDealExecutor(kmc.GaussianIso(), 128, 4, 20, 5, 1, 2.0, i + 1, rank=r, world=2, device=0) for r in range(2)]
    drv = LocalDealtEmcee(exs, 256, 4, 8)
    try:
        drv.set_positions(rng.standard_normal((256, 4)))
        drv.run(20); drv.sync(); drv.results()
    finally:
        drv.close()

def state_cycle(i):
    with kmc.Sampler(kmc.Exponential(), 256, 3, 30, 0, 1, 2.0, i + 1) as s:
        s.init_ball([0.5, 0.5, 0.5], 0.1, seed=i + 1)
        s.run(10); s.sync()
        st = s.state()
    with kmc.Sampler(kmc.Exponential(), 256, 3, 30, 0, 1, 2.0, i + 1) as t:
        t.restore(st)
        t.run(20); t.sync()

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
    ("new ExprDensity each cycle", lambda i: sampler_cycle(kmc.ExprDensity("-0.5*x*x"), 512, 8, 20)(i)),
    ("host callable density", sampler_cycle(kmc.HostLogPdf(lambda x: -0.5 * (x * x).sum(axis=1), vectorized=True), 128, 4, 10)),
    ("chain read-out both orders", lambda i: chain_cycle(i)),
    ("int_acorr", lambda i: kmc.int_acorr(rng.standard_normal((64, 128, 3)))),
    ("dealt sub-ensembles (local driver)", lambda i: dealt_cycle(i)),
    ("init_ball + state round trip", lambda i: state_cycle(i)),
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
