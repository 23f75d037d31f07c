#!/bin/bash
# N > 1 bench rehearsals on ONE GPU (gloo carries the rendezvous; RCCL refuses two ranks on one device, which exercises the
# fallback ladder): bash scripts/rehearse_ranks.sh <outdir> -- the lines land in <outdir>/bench_<n>rank_<exchange>.json
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out}
mkdir -p $OUT
export KMC_BENCH_TIMEOUT=900 KMC_BENCH_RUNG_TIMEOUT=300
for spec in "2 p2p 32768" "2 allgather 16384" "2 all 16384" "4 p2p 16384"; do
    set -- $spec; n=$1; ex=$2; w=$3
    KMC_BENCH_EXCHANGE=$ex KMC_BENCH_TEST=backend=gloo,walkers=$w python3 $R/bench.py --gpus $n --steps 2 --warmup 1 > $OUT/bench_${n}rank_${ex}.json 2> $OUT/bench_${n}rank_${ex}.err
    echo "$n ranks, exchange $ex, $w walkers per rank: rc=$?"
done
