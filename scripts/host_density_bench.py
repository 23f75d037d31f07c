"""Throughput of the host-evaluated density route (HostLogPdf / KMC_HOST_DENSITY): per half-step a
propose launch, a D2H copy, the Python callback, an H2D copy and an accept launch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import kissmcmc_jl_amd as kmc


def run(pdf, nw, nd, G, label):
    th = 0.5 + 0.1 * np.abs(np.random.default_rng(0).standard_normal((nw, nd)))
    with kmc.Sampler(pdf, nw, nd, G, 0, 1, 2.0, 1) as s:
        s.set_positions(th)
        s.run(2)
        s.sync()
        t0 = time.perf_counter()
        s.run(G)
        s.sync()
        dt = time.perf_counter() - t0
    print(f"{label:50s} {nw:6d} x {nd:4d}: {dt / (2 * G) * 1e6:9.1f} us/half-step  {nw * G / dt:12.4g} walker-steps/s")


run(kmc.HostLogPdf(lambda x: -np.inf if x < 0 else -x, scalar=True), 100, 1, 500, "python closure per walker (README density)")
run(kmc.HostLogPdf(lambda X: -0.5 * np.einsum("ij,ij->i", X, X), vectorized=True), 100, 1, 500, "numpy batch closure")
run(kmc.HostLogPdf(lambda X: -0.5 * np.einsum("ij,ij->i", X, X), vectorized=True), 4096, 32, 300, "numpy batch closure")
run(kmc.HostLogPdf(lambda X: -0.5 * np.einsum("ij,ij->i", X, X), vectorized=True), 65536, 32, 50, "numpy batch closure")
run(kmc.Exponential(), 100, 1, 4096, "menu density, resident (for scale)")
run(kmc.GaussianIso(), 65536, 32, 640, "menu density, multi-launch (for scale)")
