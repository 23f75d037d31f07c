import time, numpy as np, torch, ctypes
n = 16 * 1024 * 1024   # 128 MB
pin = torch.empty(n, dtype=torch.float64).pin_memory(); pin.fill_(1.0)
pnp = pin.numpy()
page_src = np.ones(n); dst = np.empty(n); dst[:] = 0
for name, src in (("pageable -> pageable (warm)", page_src), ("pinned (torch.pin_memory) -> pageable (warm)", pnp)):
    best = 1e9
    for r in range(3):
        t = time.perf_counter(); np.copyto(dst, src); best = min(best, time.perf_counter() - t)
    print(f"{name}: {n*8/best/1e9:.1f} GB/s")
# registered: ordinary pages pinned in place
hip = ctypes.CDLL("libamdhip64.so")
reg = np.ones(n)
rc = hip.hipHostRegister(ctypes.c_void_p(reg.ctypes.data), ctypes.c_size_t(reg.nbytes), 0)
best = 1e9
for r in range(3):
    t = time.perf_counter(); np.copyto(dst, reg); best = min(best, time.perf_counter() - t)
print(f"hipHostRegister'ed pages (rc {rc}) -> pageable (warm): {n*8/best/1e9:.1f} GB/s")
import threading
def par(src, k):
    step = n // k
    th = [threading.Thread(target=lambda i=i: np.copyto(dst[i*step:(i+1)*step], src[i*step:(i+1)*step])) for i in range(k)]
    t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; return time.perf_counter() - t
for k in (1, 2, 4, 8):
    print(f"{k} threads: pageable {n*8/min(par(page_src,k) for _ in range(3))/1e9:.1f} GB/s, pinned {n*8/min(par(pnp,k) for _ in range(3))/1e9:.1f} GB/s")
