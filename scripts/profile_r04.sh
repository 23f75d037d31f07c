#!/bin/bash
# The rocprofv3 passes behind profiles/traffic_<cfg>.json (GPU box; since round 4, hence the name): bash scripts/profile_r04.sh
#   c2/kt                 kernel trace + stats of the headline command (bench.py, C2)
#   c2/{fetch,write,l2}   PMC passes of the same command (one counter group per pass, --kernel-trace only, as the pool requires)
#   c3, c5, hbm32, hbm128 the same four passes over scripts/run_cfg.py (bench.py's other_configs shapes; hbm*: state 512 MiB)
#   probe_<cfg>.txt       in-kernel timeline of the -DKMC_PROBE build (scripts/probe_timeline.py); probe_light_<cfg>.txt: the -DKMC_PROBE=2 build (entry and last-store
#                         stamps only): body / boundary / launch period per launch, credited and burn-in phase; phase_<cfg>_m<0|1>.txt: the production build's period per phase
# The program itself follows "--" (python3 <script>), never a wrapper.  Condense HERE (the build container has git) with
# scripts/summarize_r04.py -> profiles/r04_*_summary.json and profiles/traffic_<cfg>.json.
# Started by scripts/profile_passes.sh (build container), which refuses a tree with uncommitted kernel / bench edits and leaves the commit the snapshot was taken
# from in .kmc_profile_head: the records name a commit a reader can check out (the round-4 records said "+uncommitted kernel/bench edits").
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04
if [ ! -s $R/.kmc_profile_head ]; then echo "profile_r04.sh: no .kmc_profile_head -- start the passes with scripts/profile_passes.sh (it refuses a dirty tree)"; exit 2; fi
rm -rf $OUT && mkdir -p $OUT
cp $R/.kmc_profile_head $OUT/head.txt
BENCH="python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-island"
# Round 6: every pass of a config runs in the launch mode the bench line runs that config in (the r05 C2 record was taken from the table
# graph while the line ran the updated graph): C2 and C3 = the updated graph (128 generations per replay; what their launch-mode measurement picks in every
# bench line on record), C5 and the HBM shapes = the table graph (what short jobs and C5's tie keep).  KMC_LAUNCH decides without measuring.
export KMC_LAUNCH=updated
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
for spec in "C3 1024 c3 updated" "C5 256 c5 graph" "HBM32 100 hbm32 graph" "HBM128 100 hbm128 graph"; do
    set -- $spec; cfg=$1; G=$2; lc=$3; export KMC_LAUNCH=$4
    mkdir -p $OUT/$lc
    python3 $R/scripts/run_cfg.py $cfg $G 1 > $OUT/$lc/unprofiled.txt 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$lc/kt -o t -- python3 $R/scripts/run_cfg.py $cfg $G 1 > $OUT/$lc/kt.txt 2>&1
    for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
        set -- $pass; name=$1; shift
        rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$lc/$name -o t -- python3 $R/scripts/run_cfg.py $cfg $G 1 > $OUT/$lc/$name.txt 2>&1
    done
    echo "$lc passes done"
done
# the counter CSVs of the PMC passes carry their own kernel-trace rows; drop duplicate big traces
find $OUT -path "*fetch*" -name "*kernel_trace.csv" -delete
find $OUT -path "*write*" -name "*kernel_trace.csv" -delete
find $OUT -path "*l2*" -name "*kernel_trace.csv" -delete
for spec in "C2 updated" "C3 updated" "C5 graph"; do
    set -- $spec; cfg=$1; export KMC_LAUNCH=$2
    python3 $R/scripts/probe_timeline.py $cfg > $OUT/probe_$cfg.txt 2>&1                            # eight stamps per wave: the timeline (text)
    KMC_PROBE_LIGHT=1 python3 $R/scripts/probe_timeline.py $cfg > $OUT/probe_light_$cfg.txt 2>&1    # entry + last store only: the records' body / boundary / in-kernel period
    for mom in 1 0; do python3 $R/scripts/run_cfg.py $cfg 2048 $mom > $OUT/phase_${cfg}_m$mom.txt 2>&1; done      # the production build's period per phase (credited / burn-in)
    echo "probe $cfg done"
done
unset KMC_LAUNCH
du -sh $OUT
