"""Loading of tests/golden/*.npz and the comparison used by both the CPU and the GPU leg."""
import glob
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
LOGP_RTOL = 1e-12


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def _make_golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load(name):
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    for k in ("density", "nwalkers", "ndim", "G", "nburnin", "nthin", "seed", "nmoment"):
        z[k] = int(z[k])
    z["a_scale"] = float(z["a_scale"])
    z["init"] = str(z["init"])
    z["f32"] = name.endswith("_f32")      # float rows: KMC_F32 on the device, state_f32 in the oracle
    if "theta0" not in z:   # formula-defined input (big case)
        z["theta0"] = _make_golden().theta0(z["init"], z["nwalkers"], z["ndim"], z["seed"])
    return z


def compare(z, final_pos, final_logp, naccept, msum, msumsq, nmoment, chain=None, chain_logp=None, exact_logp=False):
    np.testing.assert_array_equal(naccept, z["naccept"])
    if "final_pos" in z:
        np.testing.assert_array_equal(final_pos, z["final_pos"])
    else:
        np.testing.assert_array_equal(final_pos[:4], z["final_pos_head"])
        np.testing.assert_allclose(final_pos.sum(axis=1), z["final_pos_rowsum"], rtol=1e-13, atol=1e-12)
    if exact_logp:
        np.testing.assert_array_equal(final_logp, z["final_logp"])
    else:
        scale = np.maximum(1.0, np.abs(z["final_logp"]))
        assert np.all(np.abs(final_logp - z["final_logp"]) <= LOGP_RTOL * scale)
    assert nmoment == z["nmoment"]
    np.testing.assert_allclose(msum, z["sum"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(msumsq, z["sumsq"], rtol=1e-11, atol=1e-9)
    if chain is not None and "chain_last" in z:
        np.testing.assert_array_equal(chain[-1], z["chain_last"])
    if chain_logp is not None and "chain_logp" in z:
        scale = np.maximum(1.0, np.abs(z["chain_logp"]))
        assert np.all(np.abs(chain_logp - z["chain_logp"]) <= LOGP_RTOL * scale)
