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
    assert r["served_from"] == "infinity_cache" and "boundary" in r["limited_by"] and r["bound"] == "hbm"
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
    assert r["served_from"] == "hbm" and r["limited_by"].startswith("HBM bandwidth")
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


def test_every_fraction_of_the_committed_bench_line_follows_from_profiles():
    """scripts/recompute_roofline.py: the fractions of the newest committed bench line (headline and every other_configs entry with a
    roofline) recomputed from the algorithmic bytes, the line's own HIP-event period and the tracked records under profiles/ --
    all within 3 %."""
    import glob
    import subprocess
    import sys
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "bench_r04*.json")))[-1]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "recompute_roofline.py"), newest], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "largest deviation" in r.stdout and "DIFFERS" not in r.stdout

