"""Where does a half-step's time go?  Diagnostic build (-DKMC_PROBE) of the library: every wave of the
vector kernel stamps the 100 MHz real-time counter at entry (0), once its partner index is known (1),
once the proposal's log-pdf is reduced = both rows have arrived (2) and after its last store is issued (3).
Prints the timeline of the last generation's two launches.
Usage (GPU box): python scripts/probe_timeline.py [C2|C3|C5]   (builds libkmc_var_probe.so on first use)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "kissmcmc.jl_amd", "libkmc_var_probe.so")
os.environ["KMC_LIB_PATH"] = VAR
if not os.path.exists(VAR):
    import importlib
    b = importlib.import_module("kissmcmc.jl_amd.build".replace("kissmcmc.jl_amd", "kissmcmc_jl_amd"))
    b.build(extra_flags=["-DKMC_PROBE"], out=VAR)
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd import _lib

CONFIGS = {"C2": (kmc.GaussianIso(), 65536, 32), "C3": (kmc.Rosenbrock(), 16384, 64), "C5": (kmc.GaussianIso(), 8192, 1024),
           "C4s": (kmc.GaussianIso(), 524288, 32), "S8k": (kmc.GaussianIso(), 8192, 32)}    # S8k: one wave per CU


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    pdf, nw, nd = CONFIGS[name]
    th = np.random.default_rng(0).standard_normal((nw, nd)) * (0.1 if name == "C3" else 1.0)
    L = _lib.lib()
    for mom in (True, False):
        s = kmc.Sampler(pdf, nw, nd, 10 ** 9, 0, 1, 2.0, 7, moments=mom)
        s.set_positions(th)
        s.run(1024)
        s.sync()
        ms = s.last_run_ms()
        buf = np.zeros((2, 8192, 4), dtype=np.uint64)
        rc = L.kmc_probe_read(buf.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        print(f"== {name} moments={int(mom)}: {s.describe()}  {ms / 2048 * 1e3:.2f} us per half-step launch")
        nwave = int((buf[0, :, 0] != 0).sum())
        t = buf[:, :nwave, :].astype(np.int64)
        base = t[0, :, 0].min()
        t = (t - base) * 10.0 / 1000.0            # us since the first wave of half 0 started
        for half in (0, 1):
            a = t[half]
            print(f" half {half}: {nwave} waves; entry min/med/max {a[:, 0].min():.2f}/{np.median(a[:, 0]):.2f}/{a[:, 0].max():.2f}  "
                  f"partner-known med {np.median(a[:, 1]):.2f}  rows+logpdf med {np.median(a[:, 2]):.2f} max {a[:, 2].max():.2f}  "
                  f"end med {np.median(a[:, 3]):.2f} max {a[:, 3].max():.2f}")
            d = np.diff(a, axis=1)
            print(f"         per wave: entry->partner {np.median(d[:, 0]):.2f} (p90 {np.percentile(d[:, 0], 90):.2f}), "
                  f"partner->logpdf {np.median(d[:, 1]):.2f} (p90 {np.percentile(d[:, 1], 90):.2f}), "
                  f"logpdf->end {np.median(d[:, 2]):.2f} (p90 {np.percentile(d[:, 2], 90):.2f})")
        print(f" boundary: last wave of half 0 ends {t[0, :, 3].max():.2f}, first wave of half 1 enters {t[1, :, 0].min():.2f} "
              f"-> gap {t[1, :, 0].min() - t[0, :, 3].max():.2f} us; launch period {t[1, :, 0].min() - t[0, :, 0].min():.2f} us")
        s.close()


if __name__ == "__main__":
    main()
