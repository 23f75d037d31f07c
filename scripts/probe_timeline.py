"""Where does a half-step's time go?  Diagnostic build (-DKMC_PROBE) of the library: every wave of the
vector kernel stamps the 100 MHz real-time counter eight times (entry, Philox done, first partner loads issued, argument
struct arrived, logarithms done, rows arrived + log-pdf reduced, accept + scalar stores, last store issued) without waiting
at the stamps (kmc_kernels.hpp, KMC_STAMP).
Prints the timeline of the last generation's two launches.
Usage (GPU box): python scripts/probe_timeline.py [C2|C3|C5|R31|R63|E64]   (builds libkmc_var_probe.so on first use)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIGHT = os.environ.get("KMC_PROBE_LIGHT") == "1"        # -DKMC_PROBE=2: entry and last-store stamps only, nothing pinned in between (the records' duration source)
VAR = os.path.join(ROOT, "kissmcmc.jl_amd", "libkmc_var_probe2.so" if LIGHT else "libkmc_var_probe.so")
os.environ["KMC_LIB_PATH"] = VAR
if not os.path.exists(VAR):
    import importlib
    b = importlib.import_module("kissmcmc.jl_amd.build".replace("kissmcmc.jl_amd", "kissmcmc_jl_amd"))
    b.build(extra_flags=["-DKMC_PROBE=2" if LIGHT else "-DKMC_PROBE"], out=VAR)
import kissmcmc_jl_amd as kmc
from kissmcmc_jl_amd import _lib

CONFIGS = {"C2": (kmc.GaussianIso(), 65536, 32), "C3": (kmc.Rosenbrock(), 16384, 64), "C5": (kmc.GaussianIso(), 8192, 1024),
           "C4s": (kmc.GaussianIso(), 524288, 32), "S8k": (kmc.GaussianIso(), 8192, 32),    # S8k: one wave per CU
           "R31": (kmc.GaussianIso(), 65536, 31), "R63": (kmc.GaussianIso(), 16384, 63), "E64": (kmc.GaussianIso(), 16384, 64)}   # ragged rows against exact ones


STAMPS = ["entry", "philox done", "1st partner loads issued", "arg struct arrived", "logs done, all loads issued", "rows arrived + log-pdf",
          "accept + scalar stores", "last store issued"]


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
        s.run(1024)
        s.sync()
        ms = s.last_run_ms()
        buf = np.zeros((2, 8192, 8), dtype=np.uint64)
        how = s.describe()
        if "generation_" in how:
            # one launch per generation (kmc_generation.hpp): entry and end stamps of every workgroup's first wave, by generation parity
            rc = (L.kmc_probe_read_generation_rosenbrock if name == 'C3' else L.kmc_probe_read_generation)(buf.ctypes.data_as(ctypes.c_void_p))
            assert rc == 0
            import json
            import re
            nwg = int((buf[0, :, 0] != 0).sum())
            t = buf[:, :nwg, :].astype(np.int64)
            order = (0, 1) if t[0, :, 0].min() < t[1, :, 0].min() else (1, 0)          # the run's last two launches, earlier first
            t = (t[list(order)] - t[order[0], :, 0].min()) * 10.0 / 1000.0
            body, gap = float(t[0, :, 7].max() - t[0, :, 0].min()), float(t[1, :, 0].min() - t[0, :, 7].max())
            print(f"== {name} moments={int(mom)} KMC_LAUNCH={os.environ.get('KMC_LAUNCH', 'auto')}: {how}  {ms / 1024 * 1e3:.2f} us per generation launch ({ms / 2048 * 1e3:.2f} per half-step)")
            print(f" {nwg} workgroups stamped (first wave of each): entry spread {t[0, :, 0].max() - t[0, :, 0].min():.2f} us, wave duration median {np.median(t[0, :, 7] - t[0, :, 0]):.2f} "
                  f"(p90 {np.percentile(t[0, :, 7] - t[0, :, 0], 90):.2f}); first wave in .. last store issued {body:.2f} us; gap to the next launch's first wave {gap:.2f} us; "
                  f"launch period {t[1, :, 0].min() - t[0, :, 0].min():.2f} us")
            if "generation_group" in how and not LIGHT:
                half = nwg // 2                                  # workgroups [0, nb) carry the second half (they recompute their partner's first-half move)
                for label, sl in (("second-half workgroups", slice(0, half)), ("first-half workgroups", slice(half, nwg))):
                    dd = np.diff(t[0, sl][:, [0, 1, 2, 3, 4, 5, 7]], axis=1)
                    print(f"   {label}: entry -> Philox {np.median(dd[:, 0]):.2f}; -> rows requested {np.median(dd[:, 1]):.2f}; -> logarithms done {np.median(dd[:, 2]):.2f}; "
                          f"-> rows in, move(s) done {np.median(dd[:, 3]):.2f}; -> moments folded {np.median(dd[:, 4]):.2f}; -> last store issued {np.median(dd[:, 5]):.2f}; "
                          f"wave {np.median(t[0, sl, 7] - t[0, sl, 0]):.2f} us")
            m = re.search(r"generation_\w+[^;]*?, grid \d+ x \d+", how)
            print("PROBE_JSON " + json.dumps({"config": name, "moments": int(mom), "probe": "light" if LIGHT else "full", "execution": how, "geometry": m.group(0) if m else None, "waves_stamped": nwg, "launches_per_generation": 1,
                                              "body_us": body, "boundary_us": gap, "period_us_in_kernel": float(t[1, :, 0].min() - t[0, :, 0].min()),
                                              "period_us_hip_events_probe_build": ms / 1024 * 1e3, "wave_duration_median_us": float(np.median(t[0, :, 7] - t[0, :, 0]))}))
            s.close()
            continue
        rc = (L.kmc_probe_read_rosenbrock if name == 'C3' else L.kmc_probe_read_var if name[0] == 'R' else L.kmc_probe_read)(buf.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        print(f"== {name} moments={int(mom)} KMC_LAUNCH={os.environ.get('KMC_LAUNCH', 'auto')}: {how}  {ms / 2048 * 1e3:.2f} us per half-step launch")
        nwave = int((buf[0, :, 0] != 0).sum())
        t = buf[:, :nwave, :].astype(np.int64)
        base = t[0, :, 0].min()
        t = (t - base) * 10.0 / 1000.0            # us since the first wave of half 0 started (100 MHz counter)
        for half in (() if LIGHT else (0, 1)):
            a = t[half]
            print(f" half {half}: {nwave} waves; absolute times (min / median / max over waves):")
            for q, label in enumerate(STAMPS):
                print(f"    {q} {label:30s} {a[:, q].min():6.2f} {np.median(a[:, q]):6.2f} {a[:, q].max():6.2f}")
            d = np.diff(a, axis=1)
            print("    per wave, stage durations median (p10 .. p90): " +
                  "; ".join(f"{q}->{q + 1} {np.median(d[:, q]):.2f} ({np.percentile(d[:, q], 10):.2f}..{np.percentile(d[:, q], 90):.2f})" for q in range(7)))
            print(f"    wave duration median {np.median(a[:, 7] - a[:, 0]):.2f} (p90 {np.percentile(a[:, 7] - a[:, 0], 90):.2f})")
        print(f" boundary: last wave of half 0 ends {t[0, :, 7].max():.2f}, first wave of half 1 enters {t[1, :, 0].min():.2f} "
              f"-> gap {t[1, :, 0].min() - t[0, :, 7].max():.2f} us; launch period {t[1, :, 0].min() - t[0, :, 0].min():.2f} us")
        # machine-readable: what scripts/summarize_r04.py puts into profiles/traffic_<config>.json (body = first wave in .. last
        # store issued; boundary = the gap to the next launch's first wave; both of the -DKMC_PROBE build, whose stamps cost a little)
        import json
        import re
        m = re.search(r"half_step_\w+[^;]*?, grid \d+ x \d+", how)
        print("PROBE_JSON " + json.dumps({"config": name, "moments": int(mom), "probe": "light" if LIGHT else "full", "execution": how, "geometry": m.group(0) if m else None, "waves_stamped": nwave,
                                          "body_us": float(t[0, :, 7].max() - t[0, :, 0].min()), "boundary_us": float(t[1, :, 0].min() - t[0, :, 7].max()),
                                          "period_us_in_kernel": float(t[1, :, 0].min() - t[0, :, 0].min()), "period_us_hip_events_probe_build": ms / 2048 * 1e3,
                                          "wave_duration_median_us": float(np.median(t[0, :, 7] - t[0, :, 0]))}))
        s.close()


if __name__ == "__main__":
    main()
