"""GPU: bench.py's N > 1 path end to end -- the launcher, the self-check, the fallback ladder, the extras and the JSON fields --
with two ranks sharing ONE GPU (gloo carries the rendezvous; the peer-to-peer exchange, the sharded kernels and the dealt mode
are the same code as on two devices; RCCL itself refuses two ranks on one device, which exercises the ladder's fallbacks).
The join being distributed: reference src/samplers.jl:246-248, :273."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_env(test_opts, extra_env=None):
    """The environment of a bench.py run: without any launcher's variables, the test switches in KMC_BENCH_TEST ("opt=value,...")."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KMC_BENCH_TEST", "KMC_LAUNCH")}
    env["KMC_BENCH_TEST"] = ",".join(k if v is True else f"{k}={v}" for k, v in test_opts.items())
    env.update(extra_env or {})
    return env


def run_bench(extra_env, *args, **test_opts):
    env = bench_env({"backend": "gloo", "walkers": 4096, "timeout": 600, "rung-timeout": 240, **{k.replace("_", "-"): v for k, v in test_opts.items()}}, extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", *args],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    assert r.stdout.strip().splitlines()[-1] == lines[0]          # the result is the LAST line of stdout ...
    assert len(lines[0]) < 4096                                   # ... and fits a reader that keeps a few KB of the tail
    detail = [l for l in r.stdout.splitlines() if l.startswith('{"bench_detail"')]
    assert len(detail) == 1 and json.loads(detail[0])["bench_detail"]["value"] == json.loads(lines[0])["value"]      # the full record, one line earlier
    return json.loads(lines[0]), r.stderr


def common_checks(out):
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["warmup"] == 1
    assert out["metric"] == "walker-steps/sec" and out["scaling"] == "weak" and out["dtype"] == "f64"
    assert out["config"]["nwalkers_total"] == 2 * 4096 and out["config"]["generations"] == 1000
    assert out["check"]["timed_run_equals_unsharded_run"] is True
    assert out["check"]["nmoment"] == 2 * 4096 * 500
    assert abs(out["check"]["accept_ratio_mean"] - 0.234) < 0.01
    c = out["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["ranks_seen_by_all_reduce"] == 2
    assert c["launcher"].startswith("bench.py itself")
    assert c["native_rccl_version"] and c["native_rccl_version"].startswith("2.")
    d = out["dealt_mode"]
    assert "error" not in d, d
    assert d["value"] > 0 and d["deals"] == 1000 // 64 and abs(d["accept_ratio_mean"] - 0.234) < 0.01
    assert out["value"] == pytest.approx(2 * 4096 * 1000 / (out["ms_per_step"] * 1e-3), rel=1e-9)
    assert "roofline" in out and out["roofline"]["launches"] == 2000


def test_bench_two_ranks_p2p_from_a_plain_shell():
    out, err = run_bench({})
    common_checks(out)
    assert "peer-to-peer exchange" in out["config"]["parallelism"]
    # the link-probe rung: one "link" measured with the pull's access pattern (here: this GPU's own memory through an IPC mapping), this GPU alone
    # on its share of the walkers, and what the exact partner rule (src/samplers.jl:250) can reach from those two numbers -- in the line itself
    fab = out["fabric"]
    assert "link-probe" in [r_["rung"] for r_ in out["ladder"]] and fab["link_rate_source"] == "link-probe"
    assert fab["link_gather_GBs"] > 1.0 and fab["link_copy_GBs"] > 1.0 and fab["local_gather_GBs"] > 1.0 and fab["single_gpu_us_per_launch"] > 0.5
    assert fab["link_bound_us"] == pytest.approx(fab["bytes_per_link_per_launch"] / (fab["link_gather_GBs"] * 1e3), rel=1e-2)
    assert fab["projected_exact_speedup"] == pytest.approx(2 * fab["single_gpu_us_per_launch"] / max(fab["single_gpu_us_per_launch"], fab["link_bound_us"]), rel=1e-2)
    assert fab["ge6x_expected_under_exact_rule"] is False and fab["measured_speedup"] > 0
    assert out["repetitions"] == 3 and out["value_min"] <= out["value"] <= out["value_max"]
    assert "allgather_mode" in out                      # the north star's exchange, on record next to the pull (here: RCCL refuses one device)


def test_bench_two_ranks_allgather_ladder():
    out, err = run_bench({"KMC_BENCH_EXCHANGE": "allgather"})
    common_checks(out)
    assert "all-gather of the updated half per half-step" in out["config"]["parallelism"]
    # two ranks on one device: ncclCommInitRank refuses, every rank takes the torch-collective rung together
    assert "native RCCL all-gather set-up failed" in err
    assert "torch collective per half-step" in out["config"]["parallelism"]


def test_bench_two_ranks_under_torch_distributed_run():
    """The driver's own form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 -- bench.py is then ONE rank
    of the job and must not spawn anything itself."""
    env = bench_env({"backend": "gloo", "walkers": 4096, "no-allgather-extra": True, "rung-timeout": 240})
    from portpick import rendezvous_port
    port = rendezvous_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["check"]["timed_run_equals_unsharded_run"] is True
    assert out["collective"]["launcher"].startswith("external") and "allgather_mode" not in out
    assert "[bench launcher]" not in r.stderr
    # the peer-to-peer rung supplied `value` (not a fallback), every rank was seen, the dealt extra ran
    assert "peer-to-peer exchange" in out["config"]["parallelism"]
    assert out["collective"]["ranks_seen_by_all_reduce"] == 2 and out["collective"]["world_size"] == 2
    assert "error" not in out["dealt_mode"], out["dealt_mode"]
    assert out["dealt_mode"]["deals"] == 1000 // 64
    assert out["value_from"] == "p2p-run"
    rungs = {r_["rung"]: r_ for r_ in out["ladder"]}
    assert all(r_["ok"] for r_ in out["ladder"]), out["ladder"]
    assert {"rendezvous", "p2p-check:pull", "p2p-run", "dealt-extra"} <= set(rungs)
    assert all(r_["s"] >= 0.0 for r_ in out["ladder"])
    # a rank started by torch.distributed.run gets the same environment as one bench.py starts itself
    assert out["collective"]["rank_env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert int(out["collective"]["rank_env"]["OMP_NUM_THREADS"]) >= 1
    assert out["roofline"]["kernel"].startswith("half_step_vec<GaussianIso") and out["roofline"]["traffic"] is None


def test_bench_ladder_records_the_failed_rung():
    """A rung that fails on one rank is on record as not ok -- on every rank, rank 0's list is printed -- and `value_from` names the
    rung that supplied `value` instead."""
    out, err = run_bench({}, fault="p2p_selfcheck:0", rung_timeout=120)
    common_checks(out)
    failed = [r_ for r_ in out["ladder"] if not r_["ok"]]
    assert len(failed) >= 1 and failed[0]["rung"].startswith("p2p-check:")
    assert out["value_from"].endswith("allgather-run")


def bench_single(env_extra, launcher):
    env = bench_env({}, env_extra)
    env.pop("OMP_NUM_THREADS", None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-island"]
    r = subprocess.run([sys.executable, *launcher, *tail], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])


def test_bench_one_gpu_under_torch_distributed_run_equals_the_plain_call():
    """The N = 1 point of a scaling series launched like the N > 1 points (python -m torch.distributed.run --nproc-per-node 1: WORLD_SIZE=1,
    OMP_NUM_THREADS=1, ...) must be the same measurement as the plain `python bench.py`: same kernel and launch mode, `value` within 3 %."""
    from portpick import rendezvous_port
    for attempt in range(3):                                        # (a shared box: up to two repetitions of the pair)
        plain = bench_single({}, [])
        port = rendezvous_port()
        under = bench_single({}, ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)])
        assert under["n_gpus"] == 1 and plain["n_gpus"] == 1
        assert under["config"]["parallelism"] == plain["config"]["parallelism"] == "single GPU"
        assert under["config"]["execution"].split(" (measured")[0] == plain["config"]["execution"].split(" (measured")[0]
        assert under["roofline"]["geometry"] == plain["roofline"]["geometry"] and under["roofline"]["kernel"] == plain["roofline"]["kernel"]
        assert under["check"] == plain["check"]                     # the same job, bit for bit
        rel = abs(under["value"] / plain["value"] - 1.0)
        if rel <= 0.03:
            break
    assert rel <= 0.03, (plain["value"], under["value"])


def test_bench_sharded_ladder_over_real_rccl_with_one_rank():
    """KMC_BENCH_TEST=force-sharded: the N > 1 code path with ONE rank over the REAL collective backend ("nccl" = RCCL), which two ranks
    on one device cannot have: process group with a device id, all-reduce / all-gather-object / all_to_all_single through RCCL, and
    the native exchange -- ncclCommInitRank, the all-gathers captured into the hipGraph chunks -- inside bench.py's own ladder."""
    from portpick import rendezvous_port
    port = rendezvous_port()
    env = bench_env({"force-sharded": True, "walkers": 8192, "rung-timeout": 240}, {"MASTER_PORT": str(port)})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    c = out["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["ranks_seen_by_all_reduce"] == 1 and c["rccl_version"]
    assert out["n_gpus"] == 1 and out["check"]["timed_run_equals_unsharded_run"] is True
    assert "peer-to-peer exchange" in out["config"]["parallelism"]
    ag = out["allgather_mode"]
    assert "error" not in ag, ag
    assert ag["equals_unsharded_run"] is True and ag["captured_in_graph"] is True
    assert "error" not in out["dealt_mode"] and out["dealt_mode"]["deals"] == 1000 // 64


@pytest.mark.parametrize("point", ["p2p_setup:1", "p2p_connect:0", "p2p_selfcheck:0", "p2p_run:1"])
def test_bench_ladder_falls_through_together_when_one_rank_fails(point):
    """KMC_BENCH_TEST=fault=point:rank: ONE rank fails at a point of the peer-to-peer rung (set-up, self-check, after the timed run).  Every rank
    must then take the next rung together -- the all-gather exchange -- and the line must still be a verified result; nobody may be
    left inside a collective (the job would end in the watchdog's status 3)."""
    out, err = run_bench({}, fault=point, rung_timeout=120)
    common_checks(out)
    assert "injected fault" in err and "falling back to the RCCL all-gather exchange" in err
    assert "all-gather of the updated half per half-step" in out["config"]["parallelism"]
    assert "allgather_mode" not in out          # (the extra only accompanies the pull)


@pytest.mark.parametrize("point,key", [("dealt_setup:1", "dealt_mode"), ("allgather_setup:0", "allgather_mode")])
def test_bench_extras_fail_together_without_taking_the_result_down(point, key):
    """One rank failing in the local set-up of an EXTRA (dealt mode, all-gather record): every rank skips that extra together;
    `value` -- measured before -- stands, the extra carries an error instead of numbers, the job ends with status 0."""
    r_env = bench_env({"backend": "gloo", "walkers": 4096, "fault": point, "timeout": 600, "rung-timeout": 120})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=r_env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert out["check"]["timed_run_equals_unsharded_run"] is True and "peer-to-peer exchange" in out["config"]["parallelism"]
    assert "error" in out[key], out[key]
    other = "allgather_mode" if key == "dealt_mode" else "dealt_mode"
    if other == "dealt_mode":
        assert "error" not in out[other]


def test_bench_an_extra_that_hangs_does_not_take_the_result_down():
    """One rank never comes back from the dealt-mode extra (KMC_BENCH_TEST=fault=dealt_run_hang:1): the peers block in its collectives, the
    rung's watchdog expires on every rank -- and since `value` was measured before, rank 0 prints the line as it stands (the extra
    marked as timed out) and the job ends with status 0 instead of losing the measurement."""
    env = bench_env({"backend": "gloo", "walkers": 4096, "fault": "dealt_run_hang:1", "no-allgather-extra": True, "timeout": 600, "rung-timeout": 20})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert out["check"]["timed_run_equals_unsharded_run"] is True and "peer-to-peer exchange" in out["config"]["parallelism"]
    assert out["value"] > 0 and "did not finish within" in out["extras_timed_out"]
    assert "error" in out["dealt_mode"]
    assert out["ladder"][-1]["rung"] == "dealt-extra" and out["ladder"][-1]["timed_out"] is True
