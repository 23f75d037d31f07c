#!/bin/bash
# Is the slow mode of the HBM-resident launch a translation effect?  N fresh processes of scripts/run_cfg.py HBM32 under rocprofv3 with the address-translation
# counters (UTCL1 misses of the vector L1s, busy cycles of the shared UTCL2) and the kernel trace of the same run: duration of the half-step launches against
# the counters, process by process (the mode is a property of the process).   bash scripts/probes/hbm_tlb.sh <outdir> [processes]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/hbm_tlb}; N=${2:-8}
rm -rf $OUT; mkdir -p $OUT
for i in $(seq 1 $N); do
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p$i -o t -- python3 $R/scripts/run_cfg.py HBM32 100 0 > $OUT/p$i.txt 2>&1
done
python3 - "$OUT" "$N" <<'PY'
import csv, glob, statistics, sys
out, n = sys.argv[1], int(sys.argv[2])
print("process | us per half-step (HIP events, profiled run) | kernel-trace duration us (median of the timed half) | UTCL1 misses / requests per launch | UTCL2 busy cycles / GUI active cycles per launch")
for i in range(1, n + 1):
    f = glob.glob(f"{out}/p{i}/**/*counter_collection.csv", recursive=True)
    t = glob.glob(f"{out}/p{i}/**/*kernel_trace.csv", recursive=True)
    ev = [l for l in open(f"{out}/p{i}.txt") if "us/half-step" in l]
    us = ev[-1].split("us/half-step")[1].split()[0] if ev else "?"
    dur = "?"
    if t:
        rows = [r for r in csv.DictReader(open(t[0])) if "half_step_vec" in r["Kernel_Name"]]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][len(rows) // 2:]
        dur = f"{statistics.median(d):.2f}"
    c = {}
    if f:
        rows = [r for r in csv.DictReader(open(f[0])) if "half_step_vec" in r["Kernel_Name"]]
        for name in ("TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum", "GRBM_UTCL2_BUSY", "GRBM_GUI_ACTIVE"):
            v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name]
            c[name] = statistics.median(v[len(v) // 2:]) if v else float("nan")
    print(f"{i:7d} | {us:>8s} | {dur:>8s} | {c.get('TCP_UTCL1_TRANSLATION_MISS_sum', 0):12.0f} / {c.get('TCP_UTCL1_REQUEST_sum', 0):12.0f} | {c.get('GRBM_UTCL2_BUSY', 0):12.0f} / {c.get('GRBM_GUI_ACTIVE', 0):12.0f}")
PY
