"""GPU, -DKMC_P2P_EXPERIMENTAL library only (run by tests/test_gpu_p2p_experimental.py in a process of its own with KMC_LIB_PATH set; the
file name keeps it out of the default collection): the five exchange variants the default library does not carry -- push of accepted rows
/ lazy pull into local copies, each with the progress signal optionally folded into the half-step kernel -- must reproduce the oracle bit
for bit with every "peer" on ONE GPU.  What that cannot show is cross-GPU cache behaviour, which is why they are not in the default build.
The join being distributed: reference src/samplers.jl:246-248, :273."""
import os

import numpy as np
import pytest

import test_gpu_p2p as base
from test_gpu_p2p import G, NBURN, ND, NW, SEED, _free_port, _theta0, _worker

pytestmark = pytest.mark.gpu


def test_this_is_the_experimental_library(kmc):
    from kissmcmc_jl_amd import _lib
    assert _lib.lib().kmc_has_p2p_experimental() == 1 and "p2pexp" in _lib.LIB_PATH


def _check(oracle, tmp_path):
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NW, ND, G, NBURN, 1, 2.0, SEED), _theta0(), store_chain=False)
    z = np.load(os.path.join(str(tmp_path), "out.npz"))
    np.testing.assert_array_equal(z["nacc"], ref["naccept"])
    np.testing.assert_array_equal(z["pos"], ref["final_pos"])
    assert np.all(np.abs(z["logp"] - ref["final_logp"]) <= 1e-12 * np.maximum(1.0, np.abs(ref["final_logp"])))
    assert int(z["n"]) == ref["nmoment"]
    np.testing.assert_allclose(z["s"], ref["sum"], rtol=1e-11, atol=1e-9)


# (the parent pytest, this pytest and the ranks share the card -- 6 processes are allowed: two ranks here)
@pytest.mark.parametrize("plan", ["fold", "push", "push-fold", "lazy", "lazy-fold", ""])
def test_variant_processes_sharing_one_gpu_equal_oracle(oracle, tmp_path, plan):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), plan), nprocs=2, join=True)
    _check(oracle, tmp_path)


def test_folded_signal_tolerates_a_late_rank(oracle, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "fold", 1), nprocs=2, join=True)
    _check(oracle, tmp_path)


@pytest.mark.parametrize("kw", [dict(p2p_push=True), dict(p2p_lazy=True), dict(p2p_lazy=True, p2p_fold=True), dict(p2p_fold=True)],
                         ids=["push", "lazy", "lazy-fold", "fold"])
def test_two_variant_shards_in_one_process(kmc, oracle, kw):
    base.test_two_shards_in_one_process(kmc, oracle, kw)


def test_runtime_compiled_density_under_the_experimental_build(kmc, oracle, kmc_debug):
    """The argument struct of this build is longer (HalfStepArgs: the experimental fields sit in the middle): runtime-compiled kernels must be built with the
    same -DKMC_P2P_EXPERIMENTAL, or every field behind the first difference is read from the wrong offset (kmc_rtc.hip also compares the
    struct sizes of module and library at load)."""
    base.test_two_p2p_shards_with_a_runtime_compiled_density(kmc, oracle, "expr", kmc_debug)
    nw, nd, G = 4096, 32, 70                                # ... and an unsharded sampler over a function body
    th = np.random.default_rng(5).standard_normal((nw, nd))
    pdf = kmc.CDensity("double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;")
    with kmc.Sampler(pdf, nw, nd, G, 10, 1, 2.0, 3, moments=True) as s:
        s.set_positions(th)
        s.run(G)
        s.sync()
        pos, nacc = s.positions(), s.naccept()
    ref = oracle.emcee(oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], nw, nd, G, 10, 1, 2.0, 3), th, store_chain=False)
    np.testing.assert_array_equal(pos, ref["final_pos"])
    np.testing.assert_array_equal(nacc, ref["naccept"])
