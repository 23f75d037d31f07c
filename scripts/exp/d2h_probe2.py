import os, sys, time
import numpy as np, torch
dev = torch.device("cuda", 0)
n = 1 << 27
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
        t1 = time.perf_counter()
        st.synchronize()
        dt = time.perf_counter() - t0
    print(f"{label:34s} {n * 8 / dt / 1e9:6.1f} GB/s   enqueue {1e3*(t1-t0):7.2f} ms of {1e3*dt:7.2f} ms", flush=True)
pinned = torch.empty(n, dtype=torch.float64, pin_memory=True)
bench(pinned, "pinned (hipHostMalloc)")
arr = np.empty(n); arr[:] = 0
t = torch.from_numpy(arr)
bench(t, "pageable numpy")
print("register", torch.cuda.cudart().cudaHostRegister(arr.ctypes.data, arr.nbytes, 0))
bench(t, "registered numpy")
