#!/bin/bash
# SQ counters of the half-step / generation kernel (instruction mix, issue / wait cycles, occupancy): bash scripts/profile_sq.sh C2 [moments 0/1]
# -> gpurun_out/prof_sq_<cfg>_m<mom>/counters.json (per launch, means over the second half of the dispatches = the timed run of scripts/run_cfg.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-C2}; MOM=${2:-0}
OUT=$R/gpurun_out/prof_sq_${CFG}_m${MOM}
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/a -o t -- python3 $R/scripts/run_cfg.py $CFG 256 $MOM > $OUT/a.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/b -o t -- python3 $R/scripts/run_cfg.py $CFG 256 $MOM > $OUT/b.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/c -o t -- python3 $R/scripts/run_cfg.py $CFG 256 $MOM > $OUT/c.txt 2>&1 || echo "pass c failed"
rocprofv3 --pmc MeanOccupancyPerCU --kernel-trace --output-format csv -d $OUT/d -o t -- python3 $R/scripts/run_cfg.py $CFG 256 $MOM > $OUT/d.txt 2>&1 || echo "pass d (MeanOccupancyPerCU) failed"
python3 - <<PY
import csv, glob, collections, statistics, json
out = {"config": "$CFG", "moments": $MOM, "per": "launch (mean over the second half of the dispatches)", "counters": {}}
for sub in ("a", "b", "c", "d"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f: print("no counters in", sub); continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "half_step_" in r["Kernel_Name"] or "generation_" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["kernel"] = r["Kernel_Name"].split("(")[0]
            out["grid"], out["workgroup"] = r.get("Grid_Size"), r.get("Workgroup_Size")
            out["vgpr"], out["accum_vgpr"], out["sgpr"], out["lds"] = r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size")
    for k, v in d.items():
        out["counters"][k] = statistics.mean(v[len(v)//2:])
        print(f"$CFG m$MOM {k:24s} mean per launch {out['counters'][k]:14.1f}  (n={len(v)})")
for sub in ("a",):
    for line in open("$OUT/%s.txt" % sub):
        if "us/half-step" in line or "grid" in line: out.setdefault("run", []).append(line.strip())
json.dump(out, open("$OUT/counters.json", "w"), indent=1)
PY
