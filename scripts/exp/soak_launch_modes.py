"""Soak: the C2 job run 20 times as long (2 x 10^5 generations, 1.3 x 10^10 walker-steps per run) in each launch mode -- table
graph, updated graph (six executables taking turns), eager -- cut into uneven run() pieces: the three runs must end in the
same state bit for bit, with the same acceptance counters and moments, and the moments must be the target's.
Usage (GPU box): python scripts/exp/soak_launch_modes.py [generations]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
if len(sys.argv) > 2:
    sys.path.insert(0, ROOT)
    import kissmcmc_jl_amd as kmc
    nw, nd = 65536, 32
    th = np.random.default_rng(0).standard_normal((nw, nd))
    rng = np.random.default_rng(5)
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 10, 1, 2.0, 424242, moments=True) as s:
        s.set_positions(th)
        t = time.perf_counter()
        left = G
        while left > 0:
            n = int(min(left, rng.choice([1, 63, 64, 65, 1000, 4097, 20000])))
            s.run(n)
            left -= n
            if rng.random() < 0.2:
                s.sync()
        s.sync()
        dt = time.perf_counter() - t
        msum, msq, n = s.moments()
        np.savez(sys.argv[2], pos=s.positions(), logp=s.logp(), nacc=s.naccept(), msum=msum, msq=msq, n=n)
        mean = msum / n
        var = msq / n - mean ** 2
        print(f"KMC_LAUNCH={os.environ.get('KMC_LAUNCH'):8s} {G} generations in {dt:6.2f} s ({nw * G / dt / 1e9:5.2f} G walker-steps/s wall); accept {s.accept_ratio().mean():.4f}; "
              f"|mean|max {np.abs(mean).max():.2e}; var in [{var.min():.4f}, {var.max():.4f}]; {s.launch_count} launches", flush=True)
else:
    outs = []
    for mode in ("graph", "updated", "eager"):
        out = f"/tmp/soak_{mode}.npz"
        subprocess.run([sys.executable, os.path.abspath(__file__), str(G), out], env=dict(os.environ, KMC_LAUNCH=mode), check=True)
        outs.append(np.load(out))
    for k in ("pos", "logp", "nacc", "msum", "msq", "n"):
        assert np.array_equal(outs[0][k], outs[1][k]) and np.array_equal(outs[0][k], outs[2][k]), k
    print("the three launch modes end in the same state, counters and moments, bit for bit")
