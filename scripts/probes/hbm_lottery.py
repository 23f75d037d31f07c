"""Is the two-valued period a property of WHERE the rows were allocated?  One process, eight 512 MiB torch allocations alive at once; for each: a plain streaming
read (torch sum, GB/s), a copy into one fixed scratch buffer (GB/s), and the 2 097 152 x 32 launch with its rows bound there (moments off: nothing else of size
moves; us per half-step).  If the arenas differ inside one process, a library could allocate a few candidates and keep the fastest.   python scripts/probes/hbm_lottery.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import kissmcmc_jl_amd as kmc

NW, ND, G = 2097152, 32, 200
arenas = [torch.zeros((512 << 20) // 8, dtype=torch.float64, device="cuda") for _ in range(8)]
scratch = torch.zeros((512 << 20) // 8, dtype=torch.float64, device="cuda")


def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def period(ptr, moments=False):
    with kmc.Sampler(kmc.GaussianIso(), NW, ND, 2 * G + 64, 64, 1, 2.0, 12345, moments=moments) as s:
        s.bind_positions(ptr)
        s.init_ball(np.zeros(ND), np.ones(ND), seed=12345)
        s.run(64); s.sync()
        s.run(G); s.sync()
        return s.last_run_ms() * 1e3 / (2 * G)


for i, a in enumerate(arenas):
    rd = (512 << 20) / timed(lambda: a.sum()) / 1e9
    cp = 2 * (512 << 20) / timed(lambda: scratch.copy_(a)) / 1e9
    p1, p2 = period(a.data_ptr()), period(a.data_ptr())
    pm = period(a.data_ptr(), True)
    print(f"arena {i} at {a.data_ptr():#x}: read {rd:7.1f} GB/s, copy to scratch {cp:7.1f} GB/s, launch {p1:7.2f} {p2:7.2f} us (moments off), {pm:7.2f} (on)", flush=True)
