"""D2H bandwidth of this box by kind of host memory (torch): pinned (hipHostMalloc), page-locked in place (hipHostRegister),
pageable; 1 GiB in 128 MiB pieces on a side stream, alone and while a C2 sampler runs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kissmcmc_jl_amd as kmc

dev = torch.device("cuda", 0)
n = 1 << 27      # doubles = 1 GiB
src = torch.randn(n, dtype=torch.float64, device=dev)
st = torch.cuda.Stream(dev)
def bench(dst, label, pieces=8):
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            for i in range(pieces):
                a, b = i * n // pieces, (i + 1) * n // pieces
                dst[a:b].copy_(src[a:b], non_blocking=True)
        st.synchronize()
        dt = time.perf_counter() - t0
    print(f"{label:34s} {n * 8 / dt / 1e9:6.1f} GB/s", flush=True)

pinned = torch.empty(n, dtype=torch.float64, pin_memory=True)
bench(pinned, "pinned (hipHostMalloc)")
arr = np.empty(n)
arr[:] = 0
t = torch.from_numpy(arr)
bench(t, "pageable numpy")
rc = torch.cuda.cudart().cudaHostRegister(arr.ctypes.data, arr.nbytes, 0)
print("hipHostRegister rc", rc)
bench(t, "registered numpy (hipHostRegister)")
# while sampling
nw, nd = 65536, 32
s = kmc.Sampler(kmc.GaussianIso(), nw, nd, 10**6, 0, 1, 2.0, 1, moments=True)
s.set_positions(np.random.default_rng(0).standard_normal((nw, nd)))
s.run(2000); s.sync()
s.run(20000)
bench(pinned, "pinned, while sampling")
bench(t, "registered, while sampling")
s.sync()
print("sampler during copies:", s.last_run_ms() / 40000 * 1e3, "us per half-step")
s.close()
