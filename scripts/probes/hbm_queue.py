"""Follow-up of scripts/probes/hbm_place.py: the period of the 2 097 152 x 32 launch changes from one sampler to the next inside ONE process even when the rows sit on the
same physical pages.  What else belongs to a sampler instance: its stream (a new HIP stream per sampler: the runtime maps streams onto a small pool of hardware
queues round-robin) and its small allocations.  Here: eight samplers in a row (a) each on its own new stream, (b) all on ONE torch stream handed in with
kmc_sampler_set_stream, (c) on eight different torch streams created up front; rows bound into one arena throughout.   python scripts/probes/hbm_queue.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import kissmcmc_jl_amd as kmc

NW, ND, G = 2097152, 32, 200
arena = torch.zeros((1024 << 20) // 8, dtype=torch.float64, device="cuda")


def period(stream=None):
    with kmc.Sampler(kmc.GaussianIso(), NW, ND, 2 * G + 64, 64, 1, 2.0, 12345, moments=True) as s:
        s.bind_positions(arena.data_ptr())
        if stream is not None:
            s.set_stream(stream.cuda_stream)
        s.init_ball(np.zeros(ND), np.ones(ND), seed=12345)
        s.run(64)
        s.sync()
        s.run(G)
        s.sync()
        return s.last_run_ms() * 1e3 / (2 * G)


one = torch.cuda.Stream()
many = [torch.cuda.Stream() for _ in range(8)]
print("own new stream per sampler: " + " ".join(f"{period():7.2f}" for _ in range(8)), flush=True)
print("one shared torch stream:    " + " ".join(f"{period(one):7.2f}" for _ in range(8)), flush=True)
print("eight torch streams:        " + " ".join(f"{period(st):7.2f}" for st in many), flush=True)
print("the same eight again:       " + " ".join(f"{period(st):7.2f}" for st in many), flush=True)
