"""CPU: AddressSanitizer + UBSan over the host code that parses caller-supplied text -- the matcher for log-density bodies that stands in for the reference's
closure pdf(theta) (src/samplers.jl:257; kissmcmc.jl_amd/csrc/kmc_recognise.hpp) -- and over the oracle (scripts/sanitize_cpu.sh; GPU sanitizers are not
available on the pool).  Skipped where gcc has no sanitizer runtimes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(not (_runtime("libasan.so") and _runtime("libubsan.so")), reason="gcc has no ASan / UBSan runtime here")
def test_recogniser_and_oracle_are_clean_under_asan_and_ubsan(tmp_path):
    out = tmp_path / "sanitize.txt"
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize_cpu.sh"), str(out)], capture_output=True, text=True, timeout=600)
    text = out.read_text() if out.exists() else ""
    assert r.returncode == 0, (r.stdout + r.stderr + text)[-4000:]
    assert "recogniser: clean" in text and "asan run ok" in text and "sanitize_cpu: all clean" in text
    assert "refused 0; stateful bodies taken 0" in text               # every honest body of the grammar taken, none that carries state
    assert "ERROR: AddressSanitizer" not in text and "runtime error" not in text


def test_the_recogniser_header_needs_nothing_but_the_standard_library():
    """kmc_recognise.hpp must stay buildable by a plain host compiler (that is what lets a sanitizer see it): no HIP header, no library type."""
    src = open(os.path.join(ROOT, "kissmcmc.jl_amd", "csrc", "kmc_recognise.hpp")).read()
    includes = [l.split()[1] for l in src.splitlines() if l.startswith("#include")]
    assert includes and all(i.startswith("<") for i in includes), includes
    assert "hip" not in " ".join(includes) and "kmc_user_density" not in src
