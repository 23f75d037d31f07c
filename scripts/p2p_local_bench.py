"""The peer-to-peer exchange variants with all shards in ONE process on one GPU (kmc_sampler_p2p_connect_local): the
shards run concurrently on their own streams, ordered by the same progress flags, so kernel cost and protocol overhead
of the variants can be compared (and profiled) without processes time-slicing the card.  "Remote" rows are local here:
this measures everything except the fabric.  One host thread enqueues the shards one after the other, so G must stay
small enough for a whole run to fit the launch queues (a few thousand generations), or the first shard blocks the host
while it waits for the second.  Usage: python scripts/p2p_local_bench.py [world] [walkers_per_shard] [G] [--stats]"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per shard stream: shards spin on each other

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
G = (int(sys.argv[3]) if len(sys.argv) > 3 else 640) // 64 * 64    # whole graph chunks: eager launches of two streams of one
                                                                     # process are not reliably concurrent (seconds per run)
nw, nd = world * per, 32
th = np.random.default_rng(1).standard_normal((nw, nd))
with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 0, 1, 2.0, 9, moments=True) as s:
    s.set_positions(th)
    s.run(G)
    s.sync()
    ref_pos, ref_acc = s.positions(), s.naccept()
    print(f"unsharded {nw} x {nd}: {s.last_run_ms() / (2 * G) * 1e3:.2f} us per half-step")
variants = [("pull of drawn rows", {}), ("push of accepted rows", dict(p2p_push=True))]
for name, kw in variants:
    shards = [kmc.Sampler(kmc.GaussianIso(), nw, nd, G, 0, 1, 2.0, 9, moments=True, shard_rank=r, shard_count=world, p2p=True, **kw)
              for r in range(world)]
    if world == 2:       # streams of different priority never share a hardware queue (the shards spin on each other)
        import torch
        prio_streams = [torch.cuda.Stream(device=0, priority=-1), torch.cuda.Stream(device=0, priority=0)]
        for sh, st in zip(shards, prio_streams):
            sh.set_stream(st.cuda_stream)
    kmc.Sampler.p2p_connect_local(shards)
    best = None
    for rep in range(3):
        for sh in shards:
            sh.set_positions(th)
        t0 = time.perf_counter()
        for sh in shards:
            sh.run(G)
        for sh in shards:
            sh.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    from kissmcmc_jl_amd.distributed import local_to_global
    pos = local_to_global([sh.positions() for sh in shards], nw, world)
    ok = np.array_equal(pos, ref_pos)
    extra = ""
    print(f"{world} shards, {name:26s}: {best / (2 * G) * 1e6:6.2f} us per half-step (wall, all shards concurrent); "
          f"{'bit-identical' if ok else 'MISMATCH'}{extra}", flush=True)
    for sh in shards:
        sh.close()
