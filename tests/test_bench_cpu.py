"""bench.py's roofline block without a GPU: the tracked profile record (profiles/traffic_<cfg>.json: PMC traffic, in-kernel body /
boundary) is attached ONLY to a run that executed the very kernel geometry the record was taken from; what the line says about
where the bytes come from follows from the state size.  SURVEY 8(d)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

C2 = ("multi-launch (exact): half_step_vec L=8 K=2 ITER=2 exact-size, grid 1024 x 128, hipGraph replay of 64 generations with per-replay "
      "parameter updates (step preloaded) (measured per 64 generations: table graph 0.507 ms, updated graph 0.461 ms)")


class GaussianIso:
    pass


def test_geometry_and_kernel_name_come_from_describe():
    assert bench.kernel_geometry(C2) == "half_step_vec L=8 K=2 ITER=2 exact-size, grid 1024 x 128"
    assert bench.kernel_name(GaussianIso(), C2) == "half_step_vec<GaussianIso, L=8, K=2, ITER=2, exact-size>"
    staged = "multi-launch (exact): half_step_staged (one walker per lane, rows staged through LDS), grid 512 x 128; runtime-compiled density"
    assert bench.kernel_geometry(staged) == "half_step_staged (one walker per lane, rows staged through LDS), grid 512 x 128"
    assert bench.kernel_geometry("resident mode (exact): whole ensemble in one workgroup's LDS") is None
    gen = ("one launch per generation (exact): generation_group L=16 K=2, rows lane-striped, second-half walkers recompute their partner's first-half move, grid 2048 x 128, "
           "hipGraph replay of 64 generations")
    assert bench.kernel_geometry(gen) == "generation_group L=16 K=2, rows lane-striped, second-half walkers recompute their partner's first-half move, grid 2048 x 128"
    assert bench.kernel_name(GaussianIso(), gen) == "generation_group<GaussianIso, L=16, K=2>"
    assert bench.launches_per_generation(gen) == 1 and bench.launches_per_generation(C2) == 2
    assert bench.moment_bytes(gen) == 2048 * 128 * 2 * 8                  # the transposed fold's per-wave accumulators: 8 L / 64 doubles per thread
    # a launch of it carries both half-steps: walkers and period doubled, the fraction is the one of the half-step figures passed in
    r = bench.roofline_block(GaussianIso(), gen, 8192, 64, 3.1, 10000, bench.state_bytes(16384, 64, bench.moment_bytes(gen)), "no_such_record")
    assert r["half_steps_per_launch"] == 2 and r["avg_launch_us"] == 6.2 and r["algorithmic_read_bytes_per_launch"] == 16384 * 129 * 8
    assert abs(r["achieved"] - 8192 * 129 * 8 / 3.1e-6 / 1e9) < 1e-6 and r["traffic"] is None


def test_tracked_record_is_used_only_for_its_own_geometry():
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic_c2.json")))
    assert rec["geometry"] == bench.kernel_geometry(C2)                     # the committed record is C2's default geometry
    for key in ("kernel", "head", "hbm_bytes_per_launch", "body_us", "boundary_us", "period_us_unprofiled"):
        assert rec.get(key) is not None, key
    launch_us = 3.78
    r = bench.roofline_block(GaussianIso(), C2, 32768, 32, launch_us, 40000, bench.state_bytes(65536, 32, bench.moment_bytes(C2)), "c2")
    assert r["traffic"] == rec["hbm_bytes_per_launch"] and r["body_us"] == rec["body_us"]
    assert r["kernel"].startswith("half_step_vec<GaussianIso") and r["geometry"] == rec["geometry"]
    assert abs(r["achieved"] - 32768 * 520 / 3.78e-6 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert abs(r["body_frac"] - 32768 * 520 / (rec["body_us"] * 1e-6) / 1e9 / 8000.0) < 1e-12
    assert r["served_from"] == "infinity_cache" and r["limited_by"] == "kernel_boundary+cache_latency"
    assert r["bound"] == r["served_from"] and r["priced_against"] == "hbm_spec" and r["peak"] == 8000.0      # `bound` follows where the rows come from
    assert r["profile_record"]["record"] == "profiles/traffic_c2.json"
    assert all(not isinstance(v, str) or len(v) < 80 for v in r.values()), "numbers and short enums only: the prose lives in DESIGN.md"
    # ... and every tracked record was taken from the kernel sources of THIS tree (edit kmc_kernels.hpp / kmc_device.hpp -> profile again:
    # scripts/profile_r04.sh + scripts/summarize_r04.py)
    assert r["profile_record"]["kernel_sources_unchanged"] is True
    import glob
    for path in glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")):
        assert json.load(open(path)).get("kernel_sources_sha16") == bench.kernel_sources_sha16(), path
    # another geometry of the same workload (say, a forced plan): the record is refused, nothing of it leaks into the line
    other = C2.replace("ITER=2", "ITER=4").replace("grid 1024", "grid 512")
    r2 = bench.roofline_block(GaussianIso(), other, 32768, 32, launch_us, 40000, bench.state_bytes(65536, 32, 0), "c2")
    assert r2["traffic"] is None and r2["body_us"] is None and r2["body_frac"] is None
    assert "refused" in r2["profile_record"] and "ITER=4" in r2["profile_record"]["refused"]
    # a sharded run never takes the single-GPU record
    r3 = bench.roofline_block(GaussianIso(), C2, 32768, 32, launch_us, 40000, bench.state_bytes(65536, 32, 0), "c2", use_record=False)
    assert r3["traffic"] is None and "refused" in r3["profile_record"]


def test_a_state_beyond_the_infinity_cache_is_served_from_hbm():
    how = "multi-launch (exact): half_step_vec L=8 K=2 ITER=8 exact-size, grid 8192 x 128, hipGraph replay of 64 generations"
    sb = bench.state_bytes(2097152, 32, bench.moment_bytes(how))
    assert sb > bench.MALL_BYTES
    r = bench.roofline_block(GaussianIso(), how, 1048576, 32, 108.9, 400, sb, "hbm_2mx32")
    assert r["served_from"] == "hbm" and r["limited_by"] == "hbm_bandwidth" and r["bound"] == "hbm"
    assert abs(r["frac"] - 1048576 * 520 / 108.9e-6 / 1e9 / 8000.0) < 1e-12
    assert abs(r["frac_of_measured_copy_rate"] - r["achieved"] / 6290.0) < 1e-12
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic_hbm_2mx32.json")))
    assert rec["geometry"] == bench.kernel_geometry(how) and r["traffic"] == rec["hbm_bytes_per_launch"]
    # (algorithmic read 545 MB per launch; the counters saw 1.03 x that read and 84 MB written)
    assert 1.0 < rec["hbm_read_bytes_per_launch"] / (1048576 * 520) < 1.1


def test_test_switches_of_the_bench_live_in_one_variable(monkeypatch):
    monkeypatch.setenv("KMC_BENCH_TEST", "backend=gloo,walkers=4096,force-sharded,fault=p2p_run:1")
    assert bench.bench_test_opt("backend") == "gloo" and bench.bench_test_opt("walkers") == "4096"
    assert bench.bench_test_opt("force-sharded") is True and bench.bench_test_opt("fault") == "p2p_run:1"
    assert bench.bench_test_opt("deal-epoch", 64) == 64 and bench.bench_test_opt("no-hbm-shapes") is None


def _recompute(path, root=None):
    import importlib
    import io
    import contextlib
    sys_path = os.path.join(ROOT, "scripts")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    mod = importlib.import_module("recompute_roofline")
    old_root, old_argv = mod.ROOT, sys.argv
    buf = io.StringIO()
    try:
        mod.ROOT = root or ROOT
        sys.argv = ["recompute_roofline.py", path]
        with contextlib.redirect_stdout(buf):
            rc = mod.main()
    finally:
        mod.ROOT, sys.argv = old_root, old_argv
    return rc, buf.getvalue()


def _newest_line():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "bench_r06*.json")))[-1]


def test_every_fraction_of_the_committed_bench_line_follows_from_profiles():
    """scripts/recompute_roofline.py: the fractions of the newest committed bench line (headline and every other_configs entry with a
    roofline) recomputed from the algorithmic bytes, the line's own HIP-event period and the tracked records under profiles/ -- period-,
    duration- and body-based side by side, every record in the line's launch mode, every duration inside the time that contains it."""
    rc, out = _recompute(_newest_line())
    assert rc == 0, out[-4000:]
    assert "largest deviation" in out and "DIFFERS" not in out and "FAIL" not in out
    assert "launch mode updated_graph_128" in out and "INFLATED BY THE TOOL" in out and "duration-based" in out      # C2: the mode the bench runs; the trace's duration on record as inflated
    assert "informative" not in out                                                                                # (no figure escapes its tolerance any more)


def test_recompute_fails_on_another_launch_mode_and_on_durations_that_do_not_fit(tmp_path):
    """The three ways a record stops being evidence for a line: the line ran another launch mode than the record's passes; the record's kernel
    duration x launches exceeds the time that contains them; the rocprofv3 duration exceeds the launch period and the record does not say so."""
    import shutil
    text = open(_newest_line()).read().splitlines()
    full = json.loads(text[0])["bench_detail"]

    def write(rec, name="line.json"):
        p = tmp_path / name
        p.write_text(json.dumps({"bench_detail": rec}) + "\n")
        return str(p)

    rc, out = _recompute(write(full))
    assert rc == 0, out[-3000:]
    # (1) the same numbers from a run in the table-graph mode: the updated-graph record is refused
    other = json.loads(json.dumps(full))
    other["config"]["execution"] = other["config"]["execution"].replace("hipGraph replay of 128 generations with per-replay parameter updates (step preloaded)", "hipGraph replay of 64 generations")
    rc, out = _recompute(write(other))
    assert rc == 1 and "the record was taken in launch mode updated_graph_128, the line ran table_graph_64" in out
    # (2) a line whose launches are shorter than the record's kernel duration (ms_per_step < duration x launches)
    fast = json.loads(json.dumps(full))
    for k in ("avg_launch_us", "avg_launch_us_credited", "avg_launch_us_burnin"):
        fast["roofline"][k] = 2.5
    fast["roofline"]["achieved"] = 32768 * 520 / 2.5e-6 / 1e9
    fast["roofline"]["frac"] = fast["roofline"]["achieved"] / 8000.0
    fast["roofline"]["frac_of_measured_copy_rate"] = fast["roofline"]["achieved"] / 6290.0
    rc, out = _recompute(write(fast))
    assert rc == 1 and "does not fit the time that contains them" in out
    # (3) a record that hides an inflated trace: profiles copied, traffic_c2.json says rocprof_inflated false and takes the trace as its duration
    root = tmp_path / "root"
    shutil.copytree(os.path.join(ROOT, "profiles"), root / "profiles", ignore=shutil.ignore_patterns("bench_r0[1-5]*", "*.csv", "*.txt", "r0[1-5]*"))
    rec = json.load(open(root / "profiles" / "traffic_c2.json"))
    assert rec["rocprof_inflated"] is True and rec["launch_mode"] == "updated_graph_128" and rec["duration_source"] == "light_probe_stamps"
    rec["rocprof_inflated"], rec["duration_us"], rec["duration_source"] = False, rec["rocprof_avg_duration_us"], "rocprof_trace"
    json.dump(rec, open(root / "profiles" / "traffic_c2.json", "w"))
    rc, out = _recompute(write(full), root=str(root))
    assert rc == 1 and "exceeds the launch period" in out and "does not say so" in out


# ---- the result line must fit a reader that keeps a few KB of stdout (BENCH_r04.parsed was null, the line was 20.6 KB) ----

def _roof(name, nw, nd, us, how):
    return bench.roofline_block(GaussianIso(), how, nw // 2, nd, us, 20000, bench.state_bytes(nw, nd, bench.moment_bytes(how)), name)


def _full_single():
    """A full N = 1 record as bench.py builds it: headline + every other_configs entry (with the strings the full record carries)."""
    how3 = "multi-launch (exact): half_step_vec L=16 K=2 ITER=1 exact-size, grid 1024 x 128, hipGraph replay of 64 generations with per-replay parameter updates (step preloaded)"
    long = "x" * 400
    oc = {"C1": {"workload": long, "value": 2.08e8, "unit": "walker-steps/s", "us_per_half_step": 0.24018123456, "execution": long},
          "C3": {"workload": long, "value": 2.5452e9, "us_per_half_step": 3.2186123, "us_per_half_step_runs": [3.2, 3.3], "execution": how3, "roofline": _roof("c3", 16384, 64, 3.2186, how3)},
          "C5": {"workload": long, "value": 3.7191e8, "us_per_half_step": 11.013, "execution": long, "roofline": _roof("c5", 8192, 1024, 11.013, how3)},
          "HBM_2Mx32": {"workload": long, "value": 9.6283e9, "us_per_half_step": 108.91, "us_per_half_step_runs": [108.91234, 98.41234, 109.01234], "roofline": _roof("hbm_2mx32", 2097152, 32, 108.91, how3)},
          "HBM_512Kx128": {"workload": long, "value": 2.4467e9, "us_per_half_step": 107.14, "roofline": _roof("hbm_512kx128", 524288, 128, 107.14, how3)},
          "MID_4096x4": {"workload": long, "value": 1.2902e9, "us_per_half_step": 1.5874, "two_launches_us_per_half_step": 2.4592, "execution": long, "two_launches_execution": long},
          "MID_16384x4": {"workload": long, "value": 4.6831e9, "us_per_half_step": 1.7493, "two_launches_us_per_half_step": 2.5174, "execution": long},
          "C3_generation": {"workload": long, "value": 2.9e9, "us_per_half_step": 2.7, "two_launches_us_per_half_step": 3.22, "execution": long},
          "C2_chain_on": {"workload": long, "value": 8.9623e9, "us_per_half_step": 3.6562, "readout": long},
          "C2_user_density": {"workload": long, "value": 8.6441e9, "us_per_half_step": 3.7908, "execution": long},
          "C2_user_density_two_sums": {"error": "hiprtc: " + long},
          "C2_user_density_coupled": {"workload": long, "value": 6.4382e9, "us_per_half_step": 5.0896, "execution": long},
          "C1_user_density": {"workload": long, "value": 2.3696e8, "us_per_half_step": 0.211, "execution": long}}
    return {"metric": "walker-steps/sec", "value": 8701926704.69613, "value_min": 8655926704.123456, "value_max": 8745926704.654321, "repetitions": 3,
            "unit": "walker-steps/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 7.531205700070132,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: emcee stretch move, 65536 walkers/GPU x 32-dim isotropic Gaussian, 20000 generations (burn-in 10000), a=2, moments on, chain off",
                       "nwalkers_total": 65536, "ndim": 32, "generations": 20000, "gens_per_step": 1000, "parallelism": "single GPU", "execution": C2},
            "roofline": _roof("c2", 65536, 32, 3.763, C2),
            "check": {"accept_ratio_mean": 0.23435239, "posterior_mean_absmax": 0.0029173337, "posterior_var_min": 0.99746123, "posterior_var_max": 1.0025636, "nmoment": 655360000},
            "island_mode": {"value": 2.94856e10, "accept_ratio_mean": 0.234367, "note": long},
            "other_configs": oc,
            "cpu_baseline": {"value": 180101190.9551766, "unit": "walker-steps/s", "cores": 16, "kind": "port", "single_thread_value": 19408781.0, "thread_scaling": 9.28,
                             "sample": long, "sample_short": "C2 shape, 13383 generations = 8.77e+08 walker-steps in 4.9 s on 16 threads; 1 thread: 1852 in 6.3 s", "note": long}}


def test_the_single_gpu_result_line_fits_4_kb_and_keeps_what_the_driver_reads():
    full = _full_single()
    text = bench.compact_line(full)
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "check"):
        assert key in line, key
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["value_min"] <= line["value"] <= line["value_max"] and line["repetitions"] == 3       # >= 3 timed repetitions, the spread in the line
    assert line["other_configs"]["HBM_2Mx32"]["us_minmax"] == [98.412, 109.01]                        # (the two-valued HBM shapes carry theirs)
    r = line["roofline"]
    assert abs(r["frac"] - full["roofline"]["frac"]) < 1e-5 and r["bound"] == r["served_from"] == "infinity_cache" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["traffic"] is not None and r["body_frac"] and r["kernel"].startswith("half_step_vec<GaussianIso") and r["profile_record"]["record"] == "profiles/traffic_c2.json"
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 16 and abs(c["value"] / 180101190.9551766 - 1) < 1e-5 and "16 threads" in c["sample"]
    assert set(line["other_configs"]) == set(full["other_configs"])                       # every config is still named in the line ...
    assert line["other_configs"]["C3"]["served_from"] == "infinity_cache" and abs(line["other_configs"]["C3"]["frac"] - full["other_configs"]["C3"]["roofline"]["frac"]) < 1e-4
    assert line["other_configs"]["HBM_2Mx32"]["served_from"] == "hbm" and "error" in line["other_configs"]["C2_user_density_two_sums"]
    assert all(len(json.dumps(v)) < 160 for v in line["other_configs"].values())          # ... as numbers, not prose
    assert line["config"]["workload"].startswith("C2:") and "model" not in line["config"]


def _full_sharded(world=8, all_variants=True):
    full = {k: v for k, v in _full_single().items() if k not in ("island_mode", "other_configs", "cpu_baseline")}
    full["n_gpus"] = world
    full["config"]["parallelism"] = f"walker-sharded x{world}, exact partner rule, peer-to-peer exchange over xGMI (IPC): pull of drawn rows (system-scope loads), signal kernel; progress-flag ordering"
    full["roofline"] = bench.roofline_block(GaussianIso(), C2, 32768, 32, 14.2, 40000, bench.state_bytes(65536, 32, 0), "c2", use_record=False)
    full["check"]["timed_run_equals_unsharded_run"] = True
    full["collective"] = {"backend": "nccl", "world_size": world, "ranks_seen_by_all_reduce": world, "rccl_version": "2.26.6", "native_rccl_version": "2.26.6",
                          "launcher": "external (torch.distributed.run)", "rank_env": {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "2"}}
    tags = [v[0] for v in bench.P2P_VARIANTS] if all_variants else ["pull"]
    full["ladder"] = ([{"rung": "rendezvous", "ok": True, "s": 3.217}] + [{"rung": f"p2p-{k}:{t}", "ok": True, "s": 12.345} for t in tags for k in ("check", "time")] +
                      [{"rung": "p2p-run", "ok": True, "s": 4.5}, {"rung": "dealt-extra", "ok": True, "s": 3.1}, {"rung": "allgather-extra-setup", "ok": True, "s": 2.2},
                       {"rung": "allgather-extra", "ok": False, "s": 300.0, "timed_out": True}])
    full["value_from"] = "p2p-run"
    full["dealt_mode"] = {"value": 6.1e10, "unit": "walker-steps/s", "epoch_generations": 64, "deals": 312, "all_to_all_bytes_per_gpu_per_deal": 17825792, "accept_ratio_mean": 0.2343,
                          "posterior_mean_absmax": 0.0011, "posterior_var_minmax": [0.998, 1.002], "nmoment": 2621440000, "note": "y" * 300}
    full["allgather_mode"] = {"value": 2.3e9, "unit": "walker-steps/s", "generations": 1024, "us_per_half_step": 113.2, "equals_unsharded_run": True,
                              "bytes_received_per_gpu_per_half_step": 58720256, "execution": "RCCL all-gather of the updated half after every half-step (captured in the graph)", "note": "z" * 200}
    full["extras_timed_out"] = "'allgather-extra' did not finish within 300 s"
    full["fabric"] = {"remote_partner_bytes_per_gpu_per_launch": 7340032.0, "bytes_per_link_per_launch": 1048576.0, "link_gather_GBs": 61.234567, "link_copy_GBs": 48.123456, "local_gather_GBs": 2345.6789,
                      "link_rate_source": "link-probe", "link_bound_us": 17.123456, "single_gpu_us_per_launch": 3.7891234, "projected_exact_speedup": 1.7701234, "measured_speedup": 1.6123456,
                      "ge6x_expected_under_exact_rule": False, "push_bytes_per_link_per_launch": 1966080.0, "variants_us_per_launch": {t: 14.2 for t in tags}}
    return full


def test_the_sharded_result_line_fits_4_kb_with_every_rung_of_the_ladder():
    for world, allv in ((8, True), (2, False)):
        full = _full_sharded(world, allv)
        text = bench.compact_line(full)
        assert len(text) < 4096, len(text)
        line = json.loads(text)
        assert line["n_gpus"] == world and line["roofline"]["frac"] > 0 and line["roofline"]["traffic"] is None
        assert line["value_from"] == "p2p-run" and line["collective"]["ranks_seen_by_all_reduce"] == world and line["collective"]["launcher"].startswith("external")
        assert [r["rung"] for r in line["ladder"]] == [r["rung"] for r in full["ladder"]] and line["ladder"][-1]["timed_out"] is True
        assert line["dealt_mode"]["deals"] == 312 and line["allgather_mode"]["captured_in_graph"] is True and line["allgather_mode"]["equals_unsharded_run"] is True
        assert line["check"]["timed_run_equals_unsharded_run"] is True and "did not finish" in line["extras_timed_out"]
        # SCALE readiness: the measured link rate, the exact rule's projected speed-up next to the measured one, and ">= 6x not expected" as a field
        fab = line["fabric"]
        assert fab["link_rate_source"] == "link-probe" and fab["link_gather_GBs"] == 61.23 and fab["projected_exact_speedup"] == 1.77 and fab["measured_speedup"] == 1.612
        assert fab["ge6x_expected_under_exact_rule"] is False and "link_bound_us_at_77GBs" not in fab


def test_a_result_line_that_would_not_fit_loses_its_optional_blocks_first():
    full = _full_sharded(8, True)
    full["ladder"] = full["ladder"] * 6                  # (a pathological ladder)
    text = bench.compact_line(full)
    assert len(text) <= bench.LINE_LIMIT
    line = json.loads(text)
    assert line["value"] == full["value"] and line["roofline"]["frac"] > 0 and line["check"]["timed_run_equals_unsharded_run"] is True


def test_emit_prints_the_full_record_first_and_the_result_line_last(capsys, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _full_single()
    bench.emit(full)
    out = capsys.readouterr().out.splitlines()
    assert len(out) == 2 and out[0].startswith('{"bench_detail"') and out[1].startswith('{"metric"') and len(out[1]) < 4096
    assert json.loads(out[0])["bench_detail"]["other_configs"]["C3"]["roofline"]["geometry"] == full["other_configs"]["C3"]["roofline"]["geometry"]
    assert json.load(open(tmp_path / "bench_detail.json"))["value"] == full["value"]


def test_scale_report_reads_the_measured_fabric_block(tmp_path):
    """scripts/scale_report.py on an N > 1 line of round 6 (the link's measured rate, the exact rule's projected speed-up next to the measured one) and on a
    round-5 line (the assumed 77 GB/s, labelled as an assumption): what a reader gets from the driver's SCALE record without a GPU."""
    import subprocess
    import sys
    new = tmp_path / "n8.json"
    new.write_text(bench.compact_line(_full_sharded(8, True)) + "\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "scale_report.py"), str(new), os.path.join(ROOT, "profiles", "r05_rehearsal", "bench_2rank_drv.json")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "at 61.2 GB/s (link-probe" in r.stdout and "projected speed-up 1.77 x, measured 1.61 x; >= 6 x expected: False" in r.stdout
    assert "at an ASSUMED 77 GB/s" in r.stdout                      # (the round-5 rehearsal line)

