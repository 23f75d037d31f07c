"""One launch per generation (kmc_generation.hpp) against the two-launch kernels over ensemble shapes: us per half-step (device time,
HIP events of the second of two runs), menu Gaussian, moments on.  KMC_DEBUG=fused=1 / =0 forces either; the last column is what the
planner picks by itself.  Output: profiles/r04_generation_map.txt."""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc

NW = (2050, 4096, 8192, 16384, 32768, 65536, 131072, 262144)
ND = (1, 2, 4, 6, 8)
if "--long" in sys.argv:            # longer rows: the lane-striped form (generation_group)
    NW = (1026, 2050, 4096, 8192, 16384, 32768, 65536)
    ND = (12, 16, 32, 64, 128, 512)
store = "--chain" in sys.argv


def one(nw, nd, forced):
    if forced is None:
        os.environ.pop("KMC_DEBUG", None)
    else:
        os.environ["KMC_DEBUG"] = f"fused={forced}"
    G = int(max(128, min(20000, 2e7 / (nw * max(1, nd // 8)))))
    G -= G % 64
    with kmc.Sampler(kmc.GaussianIso(), nw, nd, 2 * G, 0, 100 if store else 1, 2.0, 3, moments=True, store_chain=store) as s:
        s.set_positions(np.random.default_rng(1).standard_normal((nw, nd)))
        s.run(G); s.sync()
        s.run(G); s.sync()
        ms = s.last_run_ms()
        how = s.describe()
    os.environ.pop("KMC_DEBUG", None)
    return 1e3 * ms / (2 * G), ("one" if "one launch per generation" in how else "two")


print("menu GaussianIso, moments on" + (", every 100th generation stored" if store else "") + ": us per half-step, two launches per generation | one launch per generation (ratio) | planner's pick")
print("walkers \\ ndim | " + " | ".join(f"{d:>32d}" for d in ND))
for nw in NW:
    cells = []
    for nd in ND:
        if nw < nd + 2:
            cells.append("-"); continue
        two, _ = one(nw, nd, 0)
        fused, how = one(nw, nd, 1)
        if how != "one":
            cells.append(f"{two:6.2f} | (no kernel)"); continue
        _, pick = one(nw, nd, None)
        cells.append(f"{two:6.2f} | {fused:6.2f} ({two / fused:4.2f}x) {pick:>4s}")
    print(f"{nw:>14d} | " + " | ".join(f"{c:>32s}" for c in cells), flush=True)
