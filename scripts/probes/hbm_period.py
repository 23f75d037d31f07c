"""The HBM-resident launches (2 097 152 x 32 and 524 288 x 128, state 512 MiB) timed in THIS process: `reps` runs of 200 generations each after 64 of warm-up,
us per half-step from HIP events -- one line per shape.  scripts/probes/hbm_bimodal.sh starts it in fresh processes (98.8 against 108.6 us box to box).
    python scripts/probes/hbm_period.py [reps] [shape: 2mx32 | 512kx128 | both]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = sys.argv[2] if len(sys.argv) > 2 else "both"
for name, nw, nd in (("2mx32", 2097152, 32), ("512kx128", 524288, 128)):
    if which not in ("both", name):
        continue
    G = 200
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, reps * G + 64, 64, 1, 2.0, 12345, moments=True) as s:
        s.init_ball(np.zeros(nd), np.ones(nd), seed=12345)
        s.run(64)
        s.sync()
        per = []
        for _ in range(reps):
            s.run(G)
            s.sync()
            per.append(s.last_run_ms() * 1e3 / (2 * G))
        print(f"{name}: us per half-step " + " ".join(f"{p:7.2f}" for p in per) + f"   | min {min(per):.2f} max {max(per):.2f} | {s.describe().split(', hipGraph')[0]}", flush=True)
