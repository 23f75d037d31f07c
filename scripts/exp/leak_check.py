"""Create / run / destroy thousands of samplers of every kind in one process: device memory, host RSS and file descriptors
before and after (leaks show as drift).  Usage (GPU box): python scripts/exp/leak_check.py [cycles]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd.metropolis import run_chains

def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0
def nfds():
    return len(os.listdir("/proc/self/fd"))
def free_mb():
    return torch.cuda.mem_get_info()[0] / 2 ** 20

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(0)
expr = kmc.ExprDensity("-0.5*x*x")
body = kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;")
kinds = [
    ("menu vec graph", lambda: (kmc.GaussianIso(), 512, 8, 70, dict(moments=True))),
    ("menu resident", lambda: (kmc.Exponential(), 100, 1, 50, dict(store_chain=True, store_logp=True))),
    ("menu odd ndim chain", lambda: (kmc.GaussianIso(), 256, 7, 70, dict(store_chain=True))),
    ("expr", lambda: (expr, 256, 16, 70, dict())),
    ("body staged", lambda: (body, 256, 16, 70, dict(moments=True))),
    ("streamed chain", lambda: (kmc.GaussianIso(), 512, 8, 200, dict(store_chain=True, store_logp=True, stream_chain=True))),
    ("streamed by walker", lambda: (kmc.GaussianIso(), 512, 8, 200, dict(store_chain=True, stream_chain=True, chain_by_walker=True))),
    ("f32", lambda: (kmc.GaussianIso(), 512, 8, 70, dict(dtype="f32"))),
    ("updated graph", lambda: (kmc.GaussianIso(), 2048, 32, 900, dict())),
]
def cycle(i):
    name, mk = kinds[i % len(kinds)]
    pdf, nw, nd, G, kw = mk()
    th = 0.6 + 0.1 * np.abs(rng.standard_normal((nw, nd))) if isinstance(pdf, kmc.Exponential) else rng.standard_normal((nw, nd))
    with kmc.Sampler(pdf, nw, nd, G, G // 4, 1, 2.0, i + 1, **kw) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        if kw.get("store_chain"):
            s.chain(logp=bool(kw.get("store_logp")), by_walker=bool(i & 1))
        s.naccept()
    if i % 11 == 0:
        run_chains(kmc.GaussianIso(), kmc.GaussianStep(0.5), rng.standard_normal((256, 3)), 40, 10, 1, i + 1)

for i in range(len(kinds) * 3):
    cycle(i)                       # warm: code objects, caches, pools
torch.cuda.synchronize()
r0, f0, d0, t0 = rss_mb(), free_mb(), nfds(), time.perf_counter()
print(f"after warm-up: RSS {r0:.0f} MiB, free device memory {f0:.0f} MiB, {d0} file descriptors", flush=True)
for i in range(N):
    cycle(i)
    if (i + 1) % 5000 == 0:
        print(f"  {i + 1} cycles: RSS {rss_mb():.0f} MiB, free {free_mb():.0f} MiB, {nfds()} fds", flush=True)
torch.cuda.synchronize()
r1, f1, d1 = rss_mb(), free_mb(), nfds()
print(f"after {N} create / run / destroy cycles ({time.perf_counter() - t0:.0f} s): RSS {r1:.0f} MiB ({r1 - r0:+.0f}), free device memory {f1:.0f} MiB ({f1 - f0:+.0f}), "
      f"{d1} file descriptors ({d1 - d0:+d})")
