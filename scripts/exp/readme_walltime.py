"""Wall time of the reference's README call (C1: 100 walkers x 1-D exponential, niter = 10^5) through kmc.emcee, repeated in
one process, by part; and the oracle (CPU restatement, one thread) beside it.
Usage (GPU box): python scripts/exp/readme_walltime.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import kissmcmc_jl_amd as kmc

pdf = kmc.Exponential()
th = kmc.make_theta0s(0.5, 0.1, pdf, 100, rng=1)
for rep in range(4):
    t = time.perf_counter()
    thetas, acc, logd, _ = kmc.emcee(pdf, th, niter=10 ** 5, use_progress_meter=False, seed=3 + rep)
    dt = time.perf_counter() - t
    print(f"kmc.emcee README call, repetition {rep}: {dt * 1e3:7.2f} ms wall  (thetas {thetas.shape}, mean {thetas.mean():.3f})", flush=True)
t = time.perf_counter()
with kmc.Sampler(pdf, 100, 1, 1000, 500, 1, 2.0, 9, store_chain=True, store_logp=True) as s:
    t1 = time.perf_counter(); s.set_positions(th.reshape(100, 1))
    t2 = time.perf_counter(); s.run(1000); s.sync()
    t3 = time.perf_counter(); a, la = s.chain(by_walker=True); ar = s.accept_ratio()
    t4 = time.perf_counter()
    dev = s.last_run_ms()
t5 = time.perf_counter()
print(f"  parts: create {1e3*(t1-t):.2f} | set_positions {1e3*(t2-t1):.2f} | run+sync {1e3*(t3-t2):.2f} (device {dev:.2f}) | read-out {1e3*(t4-t3):.2f} | destroy {1e3*(t5-t4):.2f} ms")
sys.path.insert(0, os.path.join(ROOT))
import oracle
cfg = oracle.make_config(oracle.EXPONENTIAL, [1.0], 100, 1, 1000, 500, 1, 2.0, 3, nthreads=1)
t = time.perf_counter(); r = oracle.emcee(cfg, th.reshape(100, 1)); dt = time.perf_counter() - t
print(f"CPU oracle (C restatement, 1 thread), same job: {dt * 1e3:7.2f} ms")
