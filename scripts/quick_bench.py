"""Interleaved timing of the half-step kernel at the BASELINE configs for several launch geometries
(one process, round-robin over variants; median and min reported -- cdna guide rule 24).
Usage: python scripts/quick_bench.py CONFIG [plan ...]    plan = "L,K,ITER" | "generic" | "" (default)
       env QB_MOMENTS=0/1 (default both), QB_ROUNDS (default 7), QB_GENS (default per config), QB_DTYPE=f64|f32|both"""
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissmcmc_jl_amd as kmc

CONFIGS = {
    "C2": (kmc.GaussianIso(), 65536, 32, 1024),
    "C3": (kmc.Rosenbrock(), 16384, 64, 1024),
    "C5": (kmc.GaussianIso(), 8192, 1024, 256),
    "C1": (kmc.Exponential(), 100, 1, 1024),
    "C4s": (kmc.GaussianIso(), 524288, 32, 128),     # one GPU doing the whole 8-GPU ensemble
    "S32k": (kmc.GaussianIso(), 32768, 32, 1024),
    "W128": (kmc.GaussianIso(), 65536, 128, 256),
    "W256": (kmc.GaussianIso(), 32768, 256, 256),
    "W512": (kmc.GaussianIso(), 16384, 512, 256),
    "S24k64": (kmc.GaussianIso(), 24576, 64, 1024),
}


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
    plans = sys.argv[2:] or [""]
    pdf, nw, nd, G = CONFIGS[cfgname]
    G = int(os.environ.get("QB_GENS", G))
    rounds = int(os.environ.get("QB_ROUNDS", 7))
    moms = [bool(int(os.environ["QB_MOMENTS"]))] if "QB_MOMENTS" in os.environ else [True, False]
    rng = np.random.default_rng(0)
    th = rng.standard_normal((nw, nd)) if cfgname != "C1" else 0.5 + 0.1 * np.abs(rng.standard_normal((nw, nd)))
    if cfgname == "C3":
        th *= 0.1
    variants = []
    dts = os.environ.get("QB_DTYPE", "f64")
    dts = ["f64", "f32"] if dts == "both" else [dts]
    for plan, dt in [(p, d) for p in plans for d in dts]:
        for mom in moms:
            eager = plan.endswith("e")           # "L,K,ITERe" = same geometry, eager launches (no hipGraph)
            pl = plan.rstrip("e")
            if pl:
                os.environ["KMC_PLAN"] = pl
            else:
                os.environ.pop("KMC_PLAN", None)
            s = kmc.Sampler(pdf, nw, nd, 10 ** 9, 0, 1, 2.0, 7, moments=mom, use_graph=not eager, dtype=dt)
            s.set_positions(th)
            s.run(128)
            s.sync()
            variants.append(((plan or "default") + ("/f32" if dt == "f32" else ""), mom, s, []))
    for _ in range(rounds):
        for plan, mom, s, times in variants:
            s.run(G)
            s.sync()
            times.append(s.last_run_ms())
    for plan, mom, s, times in variants:
        br = (2 * nd * (4 if plan.endswith("/f32") else 8)) + 8
        med, mn = statistics.median(times), min(times)
        acc = s.naccept().mean() / s.generation
        steps = nw * G / (med * 1e-3)
        print(f"{cfgname} plan={plan:12s} moments={int(mom)}  median {med / (2 * G) * 1e3:7.2f} us/half-step (min {mn / (2 * G) * 1e3:6.2f})  "
              f"{steps / 1e9:7.3f} Gsteps/s  read {steps * br / 1e12:6.3f} TB/s  acc={acc:.3f}", flush=True)
        s.close()


if __name__ == "__main__":
    main()
