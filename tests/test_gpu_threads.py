"""GPU: the library from several host threads at once -- one sampler per thread (the header's threading contract: any
number of handles, one thread per handle at a time; ctypes releases the GIL during a call, so the calls really overlap).
Each thread's job must equal the oracle's, whatever the others do: menu kernels, a runtime-compiled density shared by
two threads (one hiprtc cache behind a mutex), a body density, Metropolis chains, and the per-thread error string."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_samplers_from_many_threads(kmc, oracle):
    from kissmcmc_jl_amd.metropolis import run_chains
    shared = kmc.ExprDensity("-0.5*x*x")                 # compiled once per geometry, used by two threads
    body = kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) { double t = (x[i] - p[0]) * p[1]; s += t * t; } return -0.5 * s;", params=[0.0, 1.0])
    jobs = [
        ("menu gauss", kmc.GaussianIso(), oracle.GAUSSIAN_ISO, [0.0, 1.0], 2048, 32, 300),
        ("menu rosen", kmc.Rosenbrock(), oracle.ROSENBROCK, [1.0, 100.0, 20.0], 1024, 64, 200),
        ("expr a", shared, oracle.GAUSSIAN_ISO, [0.0, 1.0], 512, 16, 250),
        ("expr b", shared, oracle.GAUSSIAN_ISO, [0.0, 1.0], 512, 16, 250),
        ("body", body, oracle.GAUSSIAN_ISO, [0.0, 1.0], 640, 9, 200),
        ("body b", body, oracle.GAUSSIAN_ISO, [0.0, 1.0], 640, 9, 200),      # the same recognised body, first used by three threads at once:
        ("body c", body, oracle.GAUSSIAN_ISO, [0.0, 1.0], 512, 12, 150),     # one of them checks its per-element form, the others wait
        ("resident", kmc.Exponential(), oracle.EXPONENTIAL, [1.0], 100, 1, 1000),
    ]
    results, errors = {}, []
    barrier = threading.Barrier(len(jobs) + 2)

    def emcee_job(i, name, pdf, did, params, nw, nd, G):
        try:
            rng = np.random.default_rng(100 + i)
            th = 0.55 + 0.1 * np.abs(rng.standard_normal((nw, nd))) if did == oracle.EXPONENTIAL else 0.3 * rng.standard_normal((nw, nd))
            barrier.wait(timeout=60)
            for rep in range(3):                          # create / run / destroy repeatedly while the others do the same
                with kmc.Sampler(pdf, nw, nd, G, G // 3, 2, 2.0, 500 + i, store_chain=True, moments=True) as s:
                    s.set_positions(th)
                    s.run(G)
                    s.sync()
                    results[name] = (th, s.positions(), s.naccept(), s.chain(logp=False)[0], (did, params, nw, nd, G, 500 + i))
        except Exception as e:  # noqa: BLE001
            errors.append((name, repr(e)))

    def metropolis_job():
        try:
            barrier.wait(timeout=60)
            th = np.random.default_rng(7).standard_normal((4096, 2))
            results["metropolis"] = (th, run_chains(kmc.GaussianIso(), kmc.GaussianStep(0.8), th, 300, 100, 1, 11))
        except Exception as e:  # noqa: BLE001
            errors.append(("metropolis", repr(e)))

    def failing_job():
        """An invalid configuration in one thread: its error text must be its own (kmc_last_error is per thread)."""
        try:
            barrier.wait(timeout=60)
            for _ in range(20):
                with pytest.raises(kmc.KmcError, match="Use an even number of walkers"):
                    kmc.Sampler(kmc.GaussianIso(), 101, 4, 10)
        except Exception as e:  # noqa: BLE001
            errors.append(("failing", repr(e)))

    threads = [threading.Thread(target=emcee_job, args=(i, *job)) for i, job in enumerate(jobs)]
    threads += [threading.Thread(target=metropolis_job), threading.Thread(target=failing_job)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    for name, (th, pos, nacc, chain, (did, params, nw, nd, G, seed)) in ((k, v) for k, v in results.items() if k != "metropolis"):
        ref = oracle.emcee(oracle.make_config(did, params, nw, nd, G, G // 3, 2, 2.0, seed), th)
        np.testing.assert_array_equal(nacc, ref["naccept"], err_msg=name)
        np.testing.assert_array_equal(pos, ref["final_pos"], err_msg=name)
        np.testing.assert_array_equal(chain, ref["chain"], err_msg=name)
    th, r = results["metropolis"]
    again = run_chains(kmc.GaussianIso(), kmc.GaussianStep(0.8), th, 300, 100, 1, 11)
    np.testing.assert_array_equal(r["chain"], again["chain"])
    np.testing.assert_array_equal(r["naccept"], again["naccept"])
