"""GPU: the exchange variants that live only behind -DKMC_P2P_EXPERIMENTAL (push / lazy pull / folded signal: kmc_kernels.hpp) keep their
parity tests -- against a library built with the switch (kissmcmc.jl_amd/libkmc_var_p2pexp.so, built here when stale), in a pytest
process of its own: the product library of THIS process stays the default one.  The join being distributed: src/samplers.jl:246-248, :273."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_experimental_variants_equal_the_oracle_under_their_own_build(kmc):
    from kissmcmc_jl_amd import build as kbuild
    lib = kbuild.build_p2p_experimental()            # (hipcc: ~1 min when the sources changed since it was last built)
    env = {k: v for k, v in os.environ.items() if k not in ("KMC_PLAN", "KMC_LAUNCH")}
    env["KMC_LIB_PATH"] = lib
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "p2p_experimental_cases.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) == 13, tail
