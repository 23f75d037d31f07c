"""Where does one launch per generation stop paying?  generation_group forced on / off (KMC_DEBUG=fused=1|0) at states of 4-8 MiB, the bench's job shape
(burn-in = first half, moments on): us per half-step.   python scripts/probes/generation_limit.py   -> profiles/r05_generation_limit.txt"""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'scripts')
import kissmcmc_jl_amd as kmc
import generation_mid as gm
for name, pdf, nw, nd, scale in [("16384x64 Rosenbrock (C3)", kmc.Rosenbrock, 16384, 64, 0.1), ("16384x64 Gaussian", kmc.GaussianIso, 16384, 64, 1.0), ("32768x32 Gaussian", kmc.GaussianIso, 32768, 32, 1.0),
                                 ("8192x128 Gaussian", kmc.GaussianIso, 8192, 128, 1.0), ("65536x16 Gaussian", kmc.GaussianIso, 65536, 16, 1.0), ("12288x64 Gaussian", kmc.GaussianIso, 12288, 64, 1.0),
                                 ("24576x32 Gaussian", kmc.GaussianIso, 24576, 32, 1.0), ("16384x63 Gaussian", kmc.GaussianIso, 16384, 63, 1.0), ("32768x24 Gaussian", kmc.GaussianIso, 32768, 24, 1.0),
                                 ("16384x48 Gaussian", kmc.GaussianIso, 16384, 48, 1.0), ("4096x256 Gaussian", kmc.GaussianIso, 4096, 256, 1.0), ("2048x512 Gaussian", kmc.GaussianIso, 2048, 512, 1.0),
                                 ("20480x64 Gaussian", kmc.GaussianIso, 20480, 64, 1.0), ("40960x32 Gaussian", kmc.GaussianIso, 40960, 32, 1.0)]:
    two = gm.one(pdf, nw, nd, scale, 0)
    one = gm.one(pdf, nw, nd, scale, 1)
    pick = gm.one(pdf, nw, nd, scale, None)
    print(f"{name:26s} state {nw * (nd + nd % 2) * 8 / 2**20:5.2f} MiB | two {two[0]:6.2f} | one {one[0]:6.2f} ({two[0] / one[0]:.2f}x) | planner: {'one' if 'one launch per generation' in pick[1] else 'two'} {pick[0]:.2f} | "
          f"{one[1].split('(exact): ')[-1][:40]}", flush=True)
