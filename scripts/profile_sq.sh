#!/bin/bash
# SQ counters of the half-step kernel (instruction mix, issue / wait cycles): bash scripts/profile_sq.sh C2 [moments 0/1]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-C2}; MOM=${2:-0}
OUT=$R/gpurun_out/prof_sq_${CFG}_m${MOM}
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/a -o t -- python3 $R/scripts/run_cfg.py $CFG 256 $MOM > $OUT/a.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/b -o t -- python3 $R/scripts/run_cfg.py $CFG 256 $MOM > $OUT/b.txt 2>&1
python3 - <<PY
import csv, glob, collections, statistics
for sub in ("a", "b"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f: print("no counters in", sub); continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "half_step_vec" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        print(f"$CFG m$MOM {k:24s} mean per launch {statistics.mean(v[len(v)//2:]):14.1f}  (n={len(v)})")
PY
