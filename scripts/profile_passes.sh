#!/bin/bash
# The rocprofv3 passes behind profiles/traffic_<cfg>.json and profiles/<tag>_*_summary.json, from a COMMITTED tree (build container):
#     bash scripts/profile_passes.sh [tag, default r05]
# refuses when kernels, bench.py or the pass scripts carry uncommitted edits, records HEAD in .kmc_profile_head (git-ignored; it travels with the snapshot), runs
# scripts/profile_r04.sh on a GPU box and condenses the result here with scripts/summarize_r04.py (the records then name that commit).
set -e
cd "$(dirname "$0")/.."
tag=${1:-r06}
dirty=$(git status --porcelain -- kissmcmc.jl_amd bench.py scripts/profile_r04.sh scripts/run_cfg.py scripts/probe_timeline.py scripts/summarize_r04.py)
if [ -n "$dirty" ]; then echo "profile_passes.sh: uncommitted edits -- commit first, the records must name a commit:"; echo "$dirty"; exit 2; fi
python3 -c "from kissmcmc_jl_amd import build as b; assert not b.stale(), 'libkissmcmc_hip.so is older than its sources: build first'"
python3 -c "from kissmcmc_jl_amd import build as b; import os; b.build(extra_flags=['-DKMC_PROBE'], out=os.path.join(os.path.dirname(b.LIB), 'libkmc_var_probe.so')); b.build(extra_flags=['-DKMC_PROBE=2'], out=os.path.join(os.path.dirname(b.LIB), 'libkmc_var_probe2.so'))"   # (the probe builds of THESE sources: full timeline, light)
git rev-parse --short=12 HEAD > .kmc_profile_head
/usr/local/graft/bin/gpurun --timeout 1200 -- 'bash scripts/profile_r04.sh > gpurun_out/prof_r04.log 2>&1; tail -5 gpurun_out/prof_r04.log'
rm -f .kmc_profile_head
python3 scripts/summarize_r04.py "$tag"
