import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# runtime-compiled densities cache their code objects on disk (default: ~/.cache/kissmcmc_hip); the tests keep theirs inside the tree
os.environ.setdefault("KMC_CACHE_DIR", os.path.join(ROOT, ".kmc_cache"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """KMC_SHUFFLE_SEED=n: run the tests in a seeded random order (an order-dependent abort inside the HIP runtime was found by
    running the files in another order; the suite must not care)."""
    seed = os.environ.get("KMC_SHUFFLE_SEED")
    if seed:
        import random
        random.Random(int(seed)).shuffle(items)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    import oracle as _oracle
    _oracle.build()
    return _oracle


@pytest.fixture(scope="session")
def kmc():
    """The product package.  It refuses to import without its HIP library (no CPU fallback); in a fresh checkout the
    library is compiled first (hipcc cross-compiles gfx950 without a GPU) -- what __graft_entry__.build() does."""
    lib = os.path.join(ROOT, "kissmcmc.jl_amd", "libkissmcmc_hip.so")
    if not os.path.exists(lib) and not os.environ.get("KMC_LIB_PATH"):
        import importlib.util
        spec = importlib.util.spec_from_file_location("_kmc_build", os.path.join(ROOT, "kissmcmc.jl_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(force=True)
    import kissmcmc_jl_amd
    return kissmcmc_jl_amd


@pytest.fixture
def kmc_debug(monkeypatch):
    """The library's test-only / A-B switches live in ONE variable, KMC_DEBUG="opt[=value],opt,..." (kmc_host.hpp: debug_opt):
    `kmc_debug.set("chain-block", 1)`, `kmc_debug.unset("chain-block")`; undone with the test."""
    class _Debug:
        def __init__(self):
            # options the whole run was started with (KMC_DEBUG=poison pytest ...) stay in force; a test's own come after them and win
            self.opts = {}
            for item in filter(None, os.environ.get("KMC_DEBUG", "").split(",")):
                k, _, v = item.partition("=")
                self.opts[k] = v if v else None

        def _write(self):
            if "no-resident" in os.environ.get("KMC_DEBUG", "").split(","):       # (set by kmcenv.no_resident in the same test: keep it)
                self.opts.setdefault("no-resident", None)
            if self.opts:
                monkeypatch.setenv("KMC_DEBUG", ",".join(k if v is None else f"{k}={v}" for k, v in self.opts.items()))
            else:
                monkeypatch.delenv("KMC_DEBUG", raising=False)

        def set(self, name, value=None):
            self.opts[name] = value
            self._write()

        def unset(self, name):
            self.opts.pop(name, None)
            self._write()
    return _Debug()
