"""Host memory kept by the updated-graph launch mode (hipGraphExecKernelNodeSetParams, ~80 B per call in round 2): RSS growth of ONE sampler, segment by segment, per launch mode;
then of many short-lived samplers.   python scripts/probes/updated_graph_rss.py [segments]   (GPU box)"""
import os, sys
sys.path.insert(0, '.')
os.environ["KMC_DEBUG"] = "updated-budget-mb=100000" + ("," + os.environ["KMC_DEBUG"] if os.environ.get("KMC_DEBUG") else "")
import numpy as np
def rss():
    for l in open('/proc/self/status'):
        if l.startswith('VmRSS'): return int(l.split()[1])
import kissmcmc_jl_amd as kmc
import ctypes
_v = ctypes.c_int(); kmc._lib.lib(); ctypes.CDLL("libamdhip64.so.7").hipRuntimeGetVersion(ctypes.byref(_v))
print(f"HIP runtime in this process: {_v.value}; torch loaded: {'torch' in sys.modules}", flush=True)
NSEG = int(sys.argv[1]) if len(sys.argv) > 1 else 10
th = np.random.default_rng(1).standard_normal((65536, 32))
for launch in ("updated", "graph"):
    os.environ["KMC_LAUNCH"] = launch
    with kmc.Sampler(kmc.GaussianIso(), 65536, 32, 10 ** 9, 0, 1, 2.0, 5, moments=True) as s:
        s.set_positions(th)
        s.run(6400); s.sync()
        G = 128000; grow = []
        for seg in range(NSEG):
            r0 = rss(); s.run(G); s.sync(); grow.append(rss() - r0)
        print(f"KMC_LAUNCH={launch}: one sampler, {NSEG} segments of {G} generations ({2 * G} launches each): RSS growth per segment, KiB: {grow}; "
              f"{s.last_run_ms() * 1e3 / (2 * G):.3f} us per half-step", flush=True)
os.environ["KMC_LAUNCH"] = "updated"
grow = []
for rep in range(NSEG):
    r0 = rss()
    for i in range(100):
        with kmc.Sampler(kmc.GaussianIso(), 65536, 32, 10 ** 9, 0, 1, 2.0, 5 + i, moments=True) as s:
            s.set_positions(th); s.run(640); s.sync()
    grow.append(rss() - r0)
print(f"KMC_LAUNCH=updated: {NSEG} x 100 samplers of 640 generations each: RSS growth per 100 samplers, KiB: {grow}", flush=True)
