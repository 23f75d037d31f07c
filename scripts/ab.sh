#!/bin/bash
# A/B experiment builds of the library on one GPU box: parity tests, then quick_bench per variant.
# usage: scripts/ab.sh "<variants>" <config> <plans...>     (variant "base" = the in-tree library)
variants="$1"; shift
cfg="$1"; shift
for v in $variants; do
  if [ "$v" = "base" ]; then unset KMC_LIB_PATH; else export KMC_LIB_PATH=$PWD/kissmcmc.jl_amd/libkmc_var_$v.so; fi
  echo "=== variant $v"
  python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
  python scripts/quick_bench.py $cfg "$@" 2>&1 | grep -v amdgpu.ids
done
