#!/bin/bash
# rocprofv3 passes over the headline workload (bench.py, C2): kernel trace + stats, then the PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are collected with --kernel-trace only, as the
# GPU pool requires).  Output under gpurun_out/prof/{kt,fetch,write,l2}; condense with
#   python scripts/summarize_profile.py <tag>
# Usage (GPU box): bash scripts/profile_c2.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c2 -- python3 $R/bench.py --steps 2 --no-cpu-baseline > $OUT/kt.log 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o c2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.log 2>&1
echo "FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o c2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.log 2>&1
echo "WRITE_SIZE done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/l2 -o c2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/l2.log 2>&1
echo "L2 done"
# keep what the summary needs (the raw traces are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT/*
