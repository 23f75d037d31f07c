"""Build recipe for the HIP library (gfx950 only).  ``python -m kissmcmc_jl_amd.build``."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libkissmcmc_hip.so")
SOURCES = ["kmc_sampler.hip", "kmc_plan.hip", "kmc_launch.hip", "kmc_state.hip", "kmc_copy.hip", "kmc_rtc.hip", "kmc_p2p.hip", "kmc_metropolis_api.hip", "kmc_acorr.hip", "kmc_rccl.hip", "kmc_diag.hip", "kmc_inst_host.hip"] + \
          [f"kmc_inst_{d}{part}.hip" for part in ("", "_var", "_p2p", "_lds") for d in ("lognormal", "exponential", "gaussian_iso", "rosenbrock", "mvnormal2")]   # (longest jobs first)
HEADERS = ["kmc_host.hpp", "kmc_sampler.hpp", "kmc_device.hpp", "kmc_kernels.hpp", "kmc_islands.hpp", "kmc_generation.hpp", "kmc_copy_kernels.hpp", "kmc_metropolis.hpp", "kmc_tables.hpp", "kmc_recognise.hpp", os.path.join("..", "..", "include", "kissmcmc_hip.h")]
# kernarg preload: the half-step kernels' leading scalar parameters arrive in SGPRs at wave launch (kmc_kernels.hpp)
PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=14"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built")




def stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB, preload: bool = True) -> str:
    """Compile csrc/*.hip (one translation unit per density, in parallel) and link
    libkissmcmc_hip.so next to this file (in-tree, so it travels with the snapshot).
    ``extra_flags``/``out`` build variants side by side (``-DKMC_PROBE``)."""
    if not (force or out != LIB or stale()):
        return out
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    objdir = os.path.join(_HERE, "build_obj" + ("" if out == LIB else "_" + os.path.basename(out)))
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, *FLAGS, *(PRELOAD if preload else ()), *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    order = sorted(SOURCES, key=lambda f: not f.startswith("kmc_inst_"))       # the kernel instantiations are the long jobs: start them first
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, order))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out + ".tmp", "-lhiprtc", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    shutil.rmtree(objdir, ignore_errors=True)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
