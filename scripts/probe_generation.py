"""Where does a generation go in the one-launch-per-generation kernel?  Diagnostic build (-DKMC_PROBE) of the library: every wave of
generation_lane stamps the 100 MHz real-time counter eight times without waiting at the stamps (kmc_generation.hpp, KMC_STAMP): entry,
schedule entry arrived, my Philox block done, (second half) my partner's Philox block done, loads issued + logarithms done, rows arrived /
partner's move done, my move done, last store issued.  Prints the timeline of the run's last two launches, by half (second-half waves are
workgroups [0, nb), first-half [nb, 2 nb)).
Usage (GPU box): python scripts/probe_generation.py [walkers] [ndim]   (builds libkmc_var_probe.so on first use)"""
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
    b = importlib.import_module("kissmcmc_jl_amd.build")
    b.build(extra_flags=["-DKMC_PROBE"], out=VAR)
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd import _lib

STAMPS = ["entry", "schedule entry arrived", "my Philox block", "partner's Philox block (2nd half)", "loads issued, logarithms done",
          "rows arrived / partner's move done", "my move done", "last store issued"]


def main():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    L = _lib.lib()
    th = np.random.default_rng(0).standard_normal((nw, nd))
    for mom in (True, False):
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, 10 ** 9, 0, 1, 2.0, 7, moments=mom) as s:
            s.set_positions(th)
            s.run(1024); s.sync()
            s.run(1024); s.sync()
            ms = s.last_run_ms()
            how = s.describe()
            buf = np.zeros((2, 8192, 8), dtype=np.uint64)
            assert L.kmc_probe_read_generation(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        assert "generation_lane" in how, how
        nb = (nw // 2 + 63) // 64
        print(f"== {nw} x {nd} moments={int(mom)}: {how}  {ms / 1024 * 1e3:.2f} us per generation launch ({ms / 2048 * 1e3:.2f} per half-step)")
        t = buf[:, :2 * nb, :].astype(np.int64)
        # generation 2046 (parity 0) and 2047 (parity 1) are the run's last two launches
        base = t[0, :, 0].min()
        t = (t - base) * 10.0 / 1000.0
        for par in (0, 1):
            for name, sl in (("second-half waves", slice(0, nb)), ("first-half waves", slice(nb, 2 * nb))):
                a = t[par, sl]
                print(f" launch {par}, {name} ({a.shape[0]}): absolute times (min / median / max over waves)")
                for q, label in enumerate(STAMPS):
                    print(f"    {q} {label:36s} {a[:, q].min():6.2f} {np.median(a[:, q]):6.2f} {a[:, q].max():6.2f}")
                d = np.diff(a, axis=1)
                print("    stage durations, median: " + "; ".join(f"{q}->{q + 1} {np.median(d[:, q]):.2f}" for q in range(7)) +
                      f"; wave {np.median(a[:, 7] - a[:, 0]):.2f} (p90 {np.percentile(a[:, 7] - a[:, 0], 90):.2f})")
        body = t[0, :, 7].max() - t[0, :, 0].min()
        gap = t[1, :, 0].min() - t[0, :, 7].max()
        print(f" launch 0: first wave in .. last store issued {body:.2f} us; gap to launch 1's first wave {gap:.2f} us; period {t[1, :, 0].min() - t[0, :, 0].min():.2f} us")


if __name__ == "__main__":
    main()
