#!/bin/bash
# rocprofv3 kernel trace + stats of the one-launch-per-generation kernel (kmc_generation.hpp) and of the two-launch kernels on the same
# jobs (GPU box): bash scripts/profile_generation.sh [cfg ...] (default: MID4K MID16K; round 5: MID8Kx64 MID16Kx32 C3 -- "one" forces the kernel with
# KMC_DEBUG=fused=1 wherever it exists, C3 included).  Condensed HERE by scripts/summarize_generation.py <tag> [cfg ...] -> profiles/<tag>_generation_summary.json.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_gen
rm -rf $OUT && mkdir -p $OUT
CFGS=${@:-MID4K MID16K}
for cfg in $CFGS; do
    for mode in one two; do
        if [ $mode = two ]; then export KMC_DEBUG=fused=0; else export KMC_DEBUG=fused=1; fi
        python3 $R/scripts/run_cfg.py $cfg 4096 1 > $OUT/${cfg}_${mode}_unprofiled.txt 2>&1
        rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${cfg}_${mode} -o t -- python3 $R/scripts/run_cfg.py $cfg 4096 1 > $OUT/${cfg}_${mode}_kt.txt 2>&1
        for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do       # (PMC: one group per pass, kernel trace only)
            set -- $pass; name=$1; shift
            rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/${cfg}_${mode}_$name -o t -- python3 $R/scripts/run_cfg.py $cfg 1024 1 > $OUT/${cfg}_${mode}_$name.txt 2>&1
        done
        find $OUT -path "*_fetch*" -name "*kernel_trace.csv" -delete; find $OUT -path "*_write*" -name "*kernel_trace.csv" -delete; find $OUT -path "*_l2*" -name "*kernel_trace.csv" -delete
        echo "$cfg $mode done"
    done
done
unset KMC_DEBUG
du -sh $OUT
