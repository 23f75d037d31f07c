"""Soak of the peer-to-peer exchange variants on ONE GPU (processes share the card; IPC handles, flags and peer-mapped
rows are the ones an 8-GPU node uses): every variant x {2, 4} ranks runs G generations of an 8192 x 32 ensemble and
must end in exactly the state of the unsharded run.  Usage (GPU box): python scripts/p2p_soak.py [G] [--stats]"""
import os
import socket
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NW, ND, NBURN, SEED = 8192, 32, 100, 31337


def theta0():
    return np.random.default_rng(5).standard_normal((NW, ND))


def worker(rank, world, port, outdir, push, G):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import kissmcmc_jl_amd as kmc
    from kissmcmc_jl_amd.distributed import P2PEmcee
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        drv = P2PEmcee(kmc.GaussianIso(), NW, ND, G, NBURN, 1, 2.0, SEED, device=0, push=push)
        drv.set_positions(theta0())
        drv.run(G)
        drv.sync()
        pos, nacc = drv.positions(), drv.naccept()
        s, q, n = drv.moments()
        if rank == 0:
            np.savez(os.path.join(outdir, "out.npz"), pos=pos, nacc=nacc, s=s, n=n)
        drv.close()
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    import torch.multiprocessing as mp
    import kissmcmc_jl_amd as kmc
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    with kmc.Sampler(kmc.GaussianIso(), NW, ND, G, NBURN, 1, 2.0, SEED, moments=True) as s:
        s.set_positions(theta0())
        s.run(G)
        s.sync()
        pos, nacc, mom = s.positions(), s.naccept(), s.moments()
    bad = 0
    for world in (2, 4):
        for push in (False, True):
            with tempfile.TemporaryDirectory() as d:
                t0 = time.time()
                mp.spawn(worker, args=(world, free_port(), d, push, G), nprocs=world, join=True)
                z = np.load(os.path.join(d, "out.npz"))
                ok = (np.array_equal(z["pos"], pos) and np.array_equal(z["nacc"], nacc) and int(z["n"]) == mom[2]
                      and np.allclose(z["s"], mom[0], rtol=1e-11, atol=1e-8))
                bad += 0 if ok else 1
                print(f"{world} ranks, {'push of accepted rows' if push else 'pull of drawn rows'}: "
                      f"{G} generations {'bit-identical to the unsharded run' if ok else 'MISMATCH'}  ({time.time() - t0:.1f} s)", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
