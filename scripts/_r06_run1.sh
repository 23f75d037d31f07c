set -o pipefail
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests/test_gpu_fulljob.py -x -q -s > $O/fulljob_test.txt 2>&1; echo "fulljob rc=$?" | tee -a $O/status.txt
tail -3 $O/fulljob_test.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?" | tee -a $O/status.txt
tail -c 1500 $O/bench_a.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export KMC_LAUNCH=updated
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_upd -o t -- python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-island > $R/$O/kt_upd.json 2> $R/$O/kt_upd.err; echo "kt_upd rc=$?" | tee -a $R/$O/status.txt
unset KMC_LAUNCH
cd $R
for cfg in C2 C3 C5; do
  KMC_LAUNCH=updated KMC_PROBE_LIGHT=1 timeout -k 10 200 python scripts/probe_timeline.py $cfg > $O/probe_light_upd_$cfg.txt 2>&1; echo "light $cfg rc=$?" | tee -a $O/status.txt
done
KMC_PROBE_LIGHT=1 timeout -k 10 200 python scripts/probe_timeline.py C5 > $O/probe_light_auto_C5.txt 2>&1
KMC_LAUNCH=updated timeout -k 10 200 python scripts/probe_timeline.py C2 > $O/probe_full_upd_C2.txt 2>&1
for cfg in C2 C3 C5; do
  KMC_LAUNCH=updated timeout -k 10 100 python scripts/run_cfg.py $cfg 2048 1 > $O/runcfg_upd_$cfg.txt 2>&1
  KMC_LAUNCH=updated timeout -k 10 100 python scripts/run_cfg.py $cfg 2048 0 > $O/runcfg_upd_m0_$cfg.txt 2>&1
done
KMC_LAUNCH=updated timeout -k 10 600 bash scripts/profile_sq.sh C3 1 > $O/sq_c3.txt 2>&1; echo "sq rc=$?" | tee -a $O/status.txt
cp -r gpurun_out/prof_sq_C3_m1/counters.json $O/c3_counters_raw.json
grep PROBE_JSON $O/probe_*.txt | cut -c1-400
find $O/kt_upd -name "*kernel_trace.csv" | head
