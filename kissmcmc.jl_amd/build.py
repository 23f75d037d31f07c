"""Build recipe for the HIP library (gfx950 only).  ``python -m kissmcmc_jl_amd.build``."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libkissmcmc_hip.so")
SOURCES = ["kmc_api.hip"]
HEADERS = ["kmc_device.hpp", "kmc_kernels.hpp", os.path.join("..", "..", "include", "kissmcmc_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB) -> str:
    """Compile csrc/*.hip into libkissmcmc_hip.so next to this file (in-tree, so it travels).
    ``extra_flags``/``out`` build experiment variants (e.g. ``-DKMC_STORE_SC1``) side by side."""
    if force or out != LIB or stale():
        cmd = [_hipcc(), *FLAGS, *extra_flags, *[os.path.join(CSRC, f) for f in SOURCES], "-o", out + ".tmp",
               "-lhiprtc", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
