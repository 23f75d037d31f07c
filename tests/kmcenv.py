"""Environment switches of the library that tests flip: since round 5 they all live in KMC_DEBUG="opt[=v],..." (README)."""
import os


def _items():
    return [i for i in os.environ.get("KMC_DEBUG", "").split(",") if i]


def no_resident(monkeypatch):
    """Small ensembles in the multi-launch kernels instead of the LDS-resident one (the former KMC_NO_RESIDENT=1)."""
    items = _items()
    if "no-resident" not in items:
        monkeypatch.setenv("KMC_DEBUG", ",".join(items + ["no-resident"]))


def resident_again(monkeypatch):
    items = [i for i in _items() if i != "no-resident"]
    if items:
        monkeypatch.setenv("KMC_DEBUG", ",".join(items))
    else:
        monkeypatch.delenv("KMC_DEBUG", raising=False)
