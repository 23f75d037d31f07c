"""One launch per generation (generation_group, kmc_generation.hpp) against the two-launch kernels at the 2-8 MiB states VERDICT r04 #2 names:
C3 (16 384 x 64 chained Rosenbrock), 8 192 x 64, 32 768 x 16 and 16 384 x 32 Gaussian -- the bench's job shape (burn-in = first half, moments on),
us per half-step from HIP events of the second of two whole jobs, accept ratio beside it.  KMC_DEBUG=fused=1 / =0 forces either form.
    python scripts/generation_mid.py [G]        -> profiles/r05_generation_mid.txt"""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import kissmcmc_jl_amd as kmc

MOMENTS = '--no-moments' not in sys.argv
G = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
CASES = [("C3 16384x64 Rosenbrock", kmc.Rosenbrock, 16384, 64, 0.1), ("8192x64 Gaussian", kmc.GaussianIso, 8192, 64, 1.0),
         ("32768x16 Gaussian", kmc.GaussianIso, 32768, 16, 1.0), ("16384x32 Gaussian", kmc.GaussianIso, 16384, 32, 1.0),
         ("4096x64 Gaussian", kmc.GaussianIso, 4096, 64, 1.0), ("65536x32 Gaussian (C2)", kmc.GaussianIso, 65536, 32, 1.0)]


if "--small" in sys.argv:        # the planner's own range for the lane-striped form (state <= ~2.3 MiB)
    CASES = [(f"{nw}x{nd} Gaussian", kmc.GaussianIso, nw, nd, 1.0) for nw, nd in ((4096, 12), (16384, 12), (8192, 16), (16384, 16), (4096, 32), (8192, 32), (2050, 64), (4096, 64), (1026, 128), (2050, 128))]
    CASES += [("4096x64 Rosenbrock", kmc.Rosenbrock, 4096, 64, 0.1)]


def one(pdf, nw, nd, scale, forced, extra=""):
    if forced is None:
        os.environ.pop("KMC_DEBUG", None)
    else:
        os.environ["KMC_DEBUG"] = f"fused={forced}" + extra
    th = scale * np.random.default_rng(12345).standard_normal((nw, nd))
    with kmc.Sampler(pdf(), nw, nd, G, G // 2, 1, 2.0, 12345, moments=MOMENTS) as s:
        s.set_positions(th)
        s.run(G); s.sync()
        s.set_positions(th)
        s.run(G); s.sync()
        ms = s.last_run_ms()
        how = s.describe()
        acc = float(s.accept_ratio().mean())
        msum, msq, n = s.moments() if MOMENTS else (np.zeros(nd), None, 1)
    os.environ.pop("KMC_DEBUG", None)
    return 1e3 * ms / (2 * G), how, acc, msum / max(1, n)


if __name__ == "__main__":
    print(f"{G} generations (burn-in {G // 2}), moments {'on' if MOMENTS else 'off'}: us per half-step, two launches per generation | one launch per generation (ratio) | planner's pick")
    for name, pdf, nw, nd, scale in CASES:
        two, how2, acc2, m2 = one(pdf, nw, nd, scale, 0)
        fus, how1, acc1, m1 = one(pdf, nw, nd, scale, 1)
        _, howp, _, _ = one(pdf, nw, nd, scale, None)
        same = bool(acc1 == acc2 and np.allclose(m1, m2, rtol=1e-9, atol=1e-12))
        print(f"{name:>26s} | {two:6.2f} | {fus:6.2f} ({two / fus:4.2f}x) | pick: {'one' if 'one launch per generation' in howp else 'two'} | accept {acc1:.4f} | same accept ratio and means as the two-launch run: {same}", flush=True)
        print(f"{'':>26s}   two: {how2.split(', hipGraph')[0]}\n{'':>26s}   one: {how1.split(', hipGraph')[0]}", flush=True)
