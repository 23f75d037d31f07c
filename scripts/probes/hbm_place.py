"""Where does the two-valued period of the 2 097 152 x 32 launch come from (97.8 or 105.5-106 us per half-step, constant within a process)?  In ONE process:
  (1) a float4-style device copy of 512 MiB between fresh torch tensors, GB/s (is plain streaming two-valued too?);
  (2) the sampler with its own allocations;
  (3) the sampler's rows BOUND (kmc_sampler_bind_positions) into one 1.5 GiB torch arena at several offsets -- same physical pages of the arena every time, only the
      offset moves: 0, 2 MiB, 64 MiB, 256 MiB, 512 MiB and the odd 4 KiB + 2 MiB;
  (4) the sampler with its own allocations again, after the arena is freed.
us per half-step from HIP events (200 generations after 64 of warm-up).   python scripts/probes/hbm_place.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import kissmcmc_jl_amd as kmc

NW, ND, G = 2097152, 32, 200


def period(bind=None):
    with kmc.Sampler(kmc.GaussianIso(), NW, ND, 2 * G + 64, 64, 1, 2.0, 12345, moments=True) as s:
        if bind is not None:
            s.bind_positions(bind)
        s.init_ball(np.zeros(ND), np.ones(ND), seed=12345)
        s.run(64)
        s.sync()
        per = []
        for _ in range(2):
            s.run(G)
            s.sync()
            per.append(s.last_run_ms() * 1e3 / (2 * G))
        return min(per)


def copy_gbs():
    a = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    b = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    a.zero_(); b.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for _ in range(3):
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 10 * 2 * (512 << 20) / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best, a.data_ptr(), b.data_ptr()


out = []
g, pa, pb = copy_gbs()
out.append(f"copy 512 MiB -> 512 MiB: {g:7.1f} GB/s read+write (tensors at {pa:#x}, {pb:#x})")
out.append(f"sampler, own allocations:            {period():7.2f} us")
arena = torch.zeros((1536 << 20) // 8, dtype=torch.float64, device="cuda")
base = arena.data_ptr()
for off in (0, 2 << 20, 64 << 20, 256 << 20, 512 << 20, (2 << 20) + 4096):
    out.append(f"rows bound at arena {base:#x} + {off / 2**20:8.3f} MiB: {period(base + off):7.2f} us")
del arena
torch.cuda.empty_cache()
out.append(f"sampler, own allocations again:      {period():7.2f} us")
print(" | ".join(out), flush=True)
