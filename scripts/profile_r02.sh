#!/bin/bash
# Round-2 rocprofv3 passes (GPU box): bash scripts/profile_r02.sh [tag]
#   c2/kt      kernel trace + stats of the headline command (bench.py, C2) -> begin-to-begin period and duration per dispatch
#   c2/{fetch,write,l2}   PMC passes of the same command (one counter group per pass, --kernel-trace only, as the pool requires)
#   c3/*, c3m0/*, c5/* the same four passes over scripts/run_cfg.py C3 (streaming moments on / off) and C5 (the bench's other_configs shapes)
# The program itself follows "--" (python3 <script>), never a wrapper.  Condense with scripts/summarize_r02.py.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r02
rm -rf $OUT && mkdir -p $OUT
BENCH="python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-island"
# unprofiled reference line of the very same command, same process conditions
$BENCH > $OUT/c2_unprofiled.json 2> $OUT/c2_unprofiled.err
echo "unprofiled bench done"
mkdir -p $OUT/c2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2/kt -o t -- $BENCH > $OUT/c2/kt.json 2> $OUT/c2/kt.err
echo "c2 kernel trace done"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
    set -- $pass; name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/c2/$name -o t -- $BENCH > $OUT/c2/$name.json 2> $OUT/c2/$name.err
    echo "c2 $name done"
done
for spec in "C3 1 c3" "C3 0 c3m0" "C5 1 c5"; do
    set -- $spec; cfg=$1; mom=$2; lc=$3
    G=1024; [ $cfg = C5 ] && G=256
    mkdir -p $OUT/$lc
    python3 $R/scripts/run_cfg.py $cfg $G $mom > $OUT/$lc/unprofiled.txt 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$lc/kt -o t -- python3 $R/scripts/run_cfg.py $cfg $G $mom > $OUT/$lc/kt.txt 2>&1
    for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
        set -- $pass; name=$1; shift
        rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$lc/$name -o t -- python3 $R/scripts/run_cfg.py $cfg $G $mom > $OUT/$lc/$name.txt 2>&1
    done
    echo "$lc passes done"
done
# the counter CSVs of the PMC passes carry their own kernel-trace rows; drop duplicate big traces
find $OUT -path "*fetch*" -name "*kernel_trace.csv" -delete
find $OUT -path "*write*" -name "*kernel_trace.csv" -delete
find $OUT -path "*l2*" -name "*kernel_trace.csv" -delete
python3 $R/scripts/summarize_r02.py ${1:-r02} > $OUT/summary.txt 2>&1 || true
tail -5 $OUT/summary.txt
du -sh $OUT
