#!/bin/bash
# the HBM-resident launch (2 097 152 x 32) was 98.8 us on one box and 108.6 on another.  In ONE lease: the device's clocks, power and
# partition modes, then the same measurement in N fresh processes (fresh allocations: physical placement), clocks again in between.
#   bash scripts/probes/hbm_bimodal.sh <outdir> [processes, default 8]
out=${1:-gpurun_out/hbm_bimodal}; n=${2:-8}
mkdir -p "$out"
smi() { (rocm-smi --showclocks --showpower --showperflevel --showmemorypartition --showcomputepartition --showtemp 2>&1 | grep -v "^=\|^$" ) > "$out/smi_$1.txt"; }
smi before
rocminfo 2>/dev/null | grep -i -E "Marketing Name|Compute Unit|Max Clock|Name: +gfx" | head -8 > "$out/rocminfo.txt"
for i in $(seq 1 "$n"); do
  python3 scripts/probes/hbm_period.py 5 both 2>&1 | grep -v amdgpu.ids | sed "s/^/process $i  /"
  [ "$i" = 4 ] && smi mid
done | tee "$out/periods.txt"
smi after
# one process holding BOTH samplers' allocations in the other order (512kx128 first): does the order of allocation move the period?
python3 scripts/probes/hbm_period.py 3 512kx128 2>&1 | grep -v amdgpu.ids | sed "s/^/512kx128 alone  /" | tee -a "$out/periods.txt"
python3 scripts/probes/hbm_period.py 3 2mx32 2>&1 | grep -v amdgpu.ids | sed "s/^/2mx32 alone     /" | tee -a "$out/periods.txt"
