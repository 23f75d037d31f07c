"""The CPU oracle under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on the pool):
    gcc -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC oracle/kmc_oracle.c -o /tmp/libkmc_oracle_asan.so -lm
    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 python scripts/oracle_asan.py
(scripts/sanitize_cpu.sh does both; detect_leaks=0: the interpreter itself is not leak-clean.)
Every entry point the tests use: emcee (all menu densities, serial and OpenMP), dealt sub-ensembles with chains, the deal
permutation, the initial ball, Metropolis."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
oracle._SO = os.environ.get("KMC_ORACLE_ASAN_SO", "/tmp/libkmc_oracle_asan.so")
rng = np.random.default_rng(0)
for did, params, nw, nd in [(oracle.GAUSSIAN_ISO, [0.0, 1.0], 130, 7), (oracle.ROSENBROCK, [1.0, 100.0, 20.0], 66, 64), (oracle.EXPONENTIAL, [1.0], 100, 1),
                            (oracle.LOGNORMAL, [0.0, 1.0], 64, 3), (oracle.MVNORMAL2, [0.5, -0.25, 2.2, -0.06, 0.15], 32, 2)]:
    th = 0.6 + 0.1 * np.abs(rng.standard_normal((nw, nd)))
    for nthreads in (1, 4):
        r = oracle.emcee(oracle.make_config(did, params, nw, nd, 37, 9, 3, 2.0, 5, nthreads=nthreads), th)
        assert r["status"] == 0, (did, r["status"])
th = rng.standard_normal((256, 6))
cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], 256, 6, 40, 10, 1, 2.0, 99, nthreads=4)
r = oracle.emcee_dealt(cfg, 4, 7, th, store_chain=True); assert r["status"] == 0
r = oracle.emcee_dealt(cfg, 1, 5, th, store_chain=False); assert r["status"] == 0
print(oracle.deal_perm(7, 3, 2, 96))
r = oracle.init_ball(oracle.EXPONENTIAL, [1.0], [0.05, 0.05, 0.05], [0.5, 0.5, 0.5], 200, 3, seed=11); assert r["pos"].shape == (200, 3)
m = oracle.metropolis(oracle.GAUSSIAN_ISO, [0.0, 1.0], rng.standard_normal((50, 5)), 0.7, 60, 20, 2, 3, nthreads=2)
print("asan run ok")
