"""Large ensembles with short rows in the two-launch kernels: which geometry?  KMC_PLAN="L,K,ITER" / "generic" against the planner's
default, menu Gaussian, moments on and off; us per half-step (HIP events, second of two runs)."""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc


def one(nw, nd, plan, mom):
    if plan is None:
        os.environ.pop("KMC_PLAN", None)
    else:
        os.environ["KMC_PLAN"] = plan
    G = 512
    try:
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, 2 * G, 0, 1, 2.0, 3, moments=mom) as s:
            s.set_positions(np.random.default_rng(1).standard_normal((nw, nd)))
            s.run(G); s.sync(); s.run(G); s.sync()
            return 1e3 * s.last_run_ms() / (2 * G), s.describe().split(",")[0].replace("multi-launch (exact): ", "")
    except Exception as e:  # noqa: BLE001
        return float("nan"), str(e)[:40]
    finally:
        os.environ.pop("KMC_PLAN", None)


os.environ["KMC_DEBUG"] = "fused=0"
for nw in (65536, 262144, 1048576):
    for nd in (1, 2, 4, 8):
        chunks = (nd + 1) // 2
        plans = [None, "generic"] + [f"{L},{K},{it}" for (L, K) in ((1, 1), (2, 1), (4, 1), (4, 2), (8, 1)) if 2 * L * K >= nd for it in (1, 2, 4) if it <= L]
        cells = []
        for mom in (True, False):
            row = []
            for p in plans:
                us, how = one(nw, nd, p, mom)
                row.append(f"{p or 'default'}: {us:.2f}" + (f" [{how}]" if p is None else ""))
            cells.append(("moments on " if mom else "moments off") + " | " + " | ".join(row))
        print(f"{nw} x {nd}\n  " + "\n  ".join(cells), flush=True)
