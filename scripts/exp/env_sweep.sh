#!/bin/bash
# Which runtime knobs move the dependent-launch boundary?  C2, both launch modes, moments off / on.
# usage (GPU box): bash scripts/exp/env_sweep.sh
for setting in "BASE=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" \
               "ROC_SYSTEM_SCOPE_SIGNAL=0" "ROC_SYSTEM_SCOPE_SIGNAL=1" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "ROC_USE_FGS_KERNARG=0" "ROC_USE_FGS_KERNARG=1" \
               "DEBUG_HIP_KERNARG_COPY_OPT=0" "DEBUG_HIP_KERNARG_COPY_OPT=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=16" "DEBUG_HIP_GRAPH_BATCH_SIZE=256" \
               "AMD_DIRECT_DISPATCH=0" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1" "ROC_AQL_QUEUE_SIZE=65536" "ROC_ACTIVE_WAIT_TIMEOUT=1000"; do
  for mode in graph updated; do
    echo "== $setting KMC_LAUNCH=$mode"
    env $setting KMC_LAUNCH=$mode QB_ROUNDS=5 QB_GENS=2048 timeout -k 5 120 python scripts/quick_bench.py C2 2>&1 | grep "^C2"
  done
done
