#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_local
rm -rf $OUT && mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o p2p -- python3 $R/scripts/p2p_local_bench.py 2 65536 192 > $OUT/kt.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +30M -delete
ls $OUT/kt/*/ 2>/dev/null | head; find $OUT -name "*kernel_stats.csv" | head -2
