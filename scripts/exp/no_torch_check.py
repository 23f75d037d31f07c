"""Which HIP runtime drives the launches?  Inside a PyTorch process the library binds to the runtime the torch wheel bundles (loaded
first, same SONAME); without torch to /opt/rocm's.  The C2 job in two processes, one of each kind.
python scripts/exp/no_torch_check.py            (spawns both)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    if sys.argv[1] == "torch":
        import torch  # noqa: F401
    else:
        os.environ["KMC_DEBUG"] = "no-torch-preload"
    import numpy as np
    import kissmcmc_jl_amd as kmc
    which = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l or "libamd_comgr" in l})
    th = np.random.default_rng(0).standard_normal((65536, 32))
    for name, make in (("menu", lambda: kmc.GaussianIso()), ("C3", None)):
        if make is None:
            pdf, nw, nd, th2 = kmc.Rosenbrock(), 16384, 64, 0.1 * np.random.default_rng(0).standard_normal((16384, 64))
        else:
            pdf, nw, nd, th2 = make(), 65536, 32, th
        with kmc.Sampler(pdf, nw, nd, 10 ** 9, 0, 1, 2.0, 7, moments=True) as s:
            s.set_positions(th2)
            s.run(2048)
            s.sync()
            ts = []
            for rep in range(3):
                s.run(8192)
                s.sync()
                ts.append(s.last_run_ms() / (2 * 8192) * 1e3)
            print(f"{sys.argv[1]:8s} {name}: {min(ts):.3f} us per half-step ({', '.join(f'{t:.3f}' for t in ts)}); {s.describe()[60:140]}", flush=True)
    print("   runtime libraries:", ", ".join(which), flush=True)
else:
    for kind in ("torch", "no-torch", "torch", "no-torch"):
        subprocess.run([sys.executable, os.path.abspath(__file__), kind], check=False)
