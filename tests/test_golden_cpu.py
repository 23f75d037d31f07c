"""CPU leg: the oracle must reproduce the committed golden fixtures bit for bit (guards the
fixtures against drift of the oracle, compiler flags or libm)."""
import pytest

import goldenlib


@pytest.mark.parametrize("name", goldenlib.names())
def test_oracle_reproduces_golden(oracle, name):
    z = goldenlib.load(name)
    cfg = oracle.make_config(z["density"], list(z["params"]), z["nwalkers"], z["ndim"], z["G"], z["nburnin"],
                             z["nthin"], z["a_scale"], z["seed"], state_f32=z["f32"])
    r = oracle.emcee(cfg, z["theta0"])
    assert r["status"] == 0
    goldenlib.compare(z, r["final_pos"], r["final_logp"], r["naccept"], r["sum"], r["sumsq"], r["nmoment"],
                      r["chain"], r["chain_logp"])


def test_threaded_oracle_equals_serial(oracle):
    """The OpenMP walker loop (the 'threaded' reference path, src/samplers.jl:248) is deterministic."""
    z = goldenlib.load("gauss_256x32")
    out = []
    for nt in (1, 4):
        cfg = oracle.make_config(z["density"], list(z["params"]), z["nwalkers"], z["ndim"], z["G"], z["nburnin"],
                                 z["nthin"], z["a_scale"], z["seed"], nthreads=nt)
        out.append(oracle.emcee(cfg, z["theta0"]))
    assert (out[0]["final_pos"] == out[1]["final_pos"]).all()
    assert (out[0]["naccept"] == out[1]["naccept"]).all()
