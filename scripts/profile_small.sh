#!/bin/bash
# rocprofv3 kernel trace + stats of the small-problem paths (GPU box): bash scripts/profile_small.sh
#   readme   the reference's README call through the drop-in surface, niter 10^5 .. 10^8 (resident kernel + its draw table)
#   metro    one Metropolis chain / a few / many (table kernel + its draw fill, in-kernel draws beyond 16 384 chains)
# The program itself follows "--" (python3 <script>), never a wrapper.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$R
OUT=$R/gpurun_out/prof_small
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/readme -o t -- python3 $R/scripts/readme_walltime.py > $OUT/readme.txt 2>&1
echo "readme done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/metro -o t -- python3 $R/scripts/metropolis_single_chain.py > $OUT/metro.txt 2>&1
echo "metro done"
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -30
