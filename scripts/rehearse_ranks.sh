#!/bin/bash
# N > 1 bench rehearsals on ONE GPU (gloo carries the rendezvous; RCCL refuses two ranks on one device, which exercises the
# fallback ladder): bash scripts/rehearse_ranks.sh <outdir> -- the lines land in <outdir>/bench_<n>rank_<exchange>.json.
# The "drv" lines are launched the way the driver launches a scaling point (python -m torch.distributed.run --nnodes=1
# --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...: bench.py is then ONE rank); the others from a
# plain shell (bench.py creates its ranks).  scripts/scale_report.py reads the drv lines at the end.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out}
mkdir -p $OUT
for spec in "2 p2p 32768" "2 allgather 16384"; do      # (four ranks on one GPU time-slice it while every workgroup spin-waits on its peers: profiles/r05_rehearsal/scale_report_drv.txt is that record, annotated)
    set -- $spec; n=$1; ex=$2; w=$3
    KMC_BENCH_EXCHANGE=$ex KMC_BENCH_TEST=backend=gloo,walkers=$w,timeout=900,rung-timeout=300 python3 $R/bench.py --gpus $n --steps 2 --warmup 1 > $OUT/bench_${n}rank_${ex}.json 2> $OUT/bench_${n}rank_${ex}.err
    echo "$n ranks, exchange $ex, $w walkers per rank: rc=$?"
done
port=23450
for spec in "1 65536" "2 65536"; do
    set -- $spec; n=$1; w=$2; port=$((port + 7))
    KMC_BENCH_TEST=backend=gloo,walkers=$w,no-hbm-shapes,timeout=900,rung-timeout=300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        $R/bench.py --gpus $n --steps 4 --warmup 1 > $OUT/bench_${n}rank_drv.json 2> $OUT/bench_${n}rank_drv.err
    echo "$n ranks under torch.distributed.run, $w walkers per rank: rc=$?"
done
python3 $R/scripts/scale_report.py $OUT/bench_1rank_drv.json $OUT/bench_2rank_drv.json > $OUT/scale_report_drv.txt 2>&1
cat $OUT/scale_report_drv.txt
