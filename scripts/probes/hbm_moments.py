"""Follow-up of scripts/probes/hbm_queue.py (neither the rows' pages nor the stream): eight samplers in a row over the same bound rows, moments off / on, and with the
library's device-buffer cache emptied between samplers (fresh small allocations every time).   python scripts/probes/hbm_moments.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd import _lib

NW, ND, G = 2097152, 32, 200
arena = torch.zeros((1024 << 20) // 8, dtype=torch.float64, device="cuda")


def period(moments, release=False):
    if release:
        _lib.lib().kmc_device_cache_release()
    with kmc.Sampler(kmc.GaussianIso(), NW, ND, 2 * G + 64, 64, 1, 2.0, 12345, moments=moments) as s:
        s.bind_positions(arena.data_ptr())
        s.init_ball(np.zeros(ND), np.ones(ND), seed=12345)
        s.run(64)
        s.sync()
        s.run(G)
        s.sync()
        return s.last_run_ms() * 1e3 / (2 * G)


print("moments off:                     " + " ".join(f"{period(False):7.2f}" for _ in range(8)), flush=True)
print("moments on:                      " + " ".join(f"{period(True):7.2f}" for _ in range(8)), flush=True)
print("moments on, cache emptied first: " + " ".join(f"{period(True, True):7.2f}" for _ in range(8)), flush=True)
print("moments off again:               " + " ".join(f"{period(False):7.2f}" for _ in range(8)), flush=True)
