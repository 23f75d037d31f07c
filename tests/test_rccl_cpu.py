"""CPU: the native RCCL route's readiness without hardware (the all-gather of the updated half is the join of
reference src/samplers.jl:273 across GPUs).  Two ranks cannot share one device under RCCL, so what CAN be checked without a
node is checked here: the library resolves every RCCL entry point it uses from the image's librccl.so, the unique id
round-trips through the C ABI, and bench.py's own launcher ends a failing job with a non-zero status instead of hanging."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_entry_points_resolve_and_report_a_version(kmc):
    from kissmcmc_jl_amd import _lib
    v, path = C.c_int(0), C.create_string_buffer(1024)
    assert _lib.lib().kmc_rccl_version(C.byref(v), path, 1024) == _lib.OK, _lib.lib().kmc_last_error().decode()
    # KMC_OK means: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString, ncclGetVersion all resolved
    assert v.value >= 22000, v.value                       # RCCL 2.20+ (graph capture of collectives)
    assert os.path.basename(path.value.decode()).startswith("librccl.so")
    assert kmc.Sampler.rccl_version() == f"{v.value // 10000}.{v.value // 100 % 100}.{v.value % 100}"


def test_rccl_unique_id_round_trips(kmc):
    from kissmcmc_jl_amd import _lib
    a, b = kmc.Sampler.rccl_unique_id(), kmc.Sampler.rccl_unique_id()
    assert len(a) == len(b) == _lib.RCCL_ID_BYTES == 128
    assert any(a) and a != b                               # a real id, a fresh one per call
    # the blob is what travels between ranks (pickled by torch.distributed, or any other transport): bytes in, same bytes out
    buf = C.create_string_buffer(bytes(a), _lib.RCCL_ID_BYTES)
    assert buf.raw == a
    assert _lib.lib().kmc_rccl_unique_id(None) == _lib.ERR_BAD_ARG
    assert _lib.lib().kmc_sampler_rccl_init(None, buf) == _lib.ERR_BAD_ARG
    got = C.c_int(7)
    assert _lib.lib().kmc_sampler_rccl_capture(None, C.byref(got)) == _lib.ERR_BAD_ARG


def test_bench_launcher_starts_its_own_ranks_and_relays_failure():
    """`python3 bench.py --gpus 2` from a plain shell: the launcher creates the two ranks itself (no torch.distributed.run
    around it).  Without a GPU both ranks refuse to run; the launcher must come back promptly with a non-zero status and the
    ranks' message -- never hang, never report success without a result line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["KMC_BENCH_TEST"] = "timeout=240"
    try:
        import torch
        if torch.cuda.is_available():
            env["HIP_VISIBLE_DEVICES"] = ""                 # (on a GPU box: hide the device, the launcher's failure path is the subject)
            env["CUDA_VISIBLE_DEVICES"] = ""
    except ImportError:
        pass
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr
    assert "a rank exited with status" in r.stderr or r.stderr.count("needs an MI355X") >= 2
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]


def test_bench_launcher_is_not_used_under_a_launcher():
    """With WORLD_SIZE in the environment (torch.distributed.run, the driver's form) bench.py is ONE rank: it must not spawn."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577",
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs an MI355X" in r.stderr and "[bench launcher]" not in r.stderr


def test_bench_launcher_takes_its_ranks_down_when_it_is_stopped():
    """The ranks are sessions of their own; a launcher that is told to stop (the harness's timeout sends SIGTERM) must end them -- not
    leave them on the GPUs as orphans -- and exit non-zero.  (--test-sleep: every rank just sleeps; no GPU involved.)"""
    import re
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--test-sleep", "120"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = ""
    t_end = time.monotonic() + 60
    while "started ranks:" not in line and time.monotonic() < t_end:
        line = p.stderr.readline()
    pids = [int(v) for v in re.findall(r"\d+", line.split("started ranks:")[1])]
    assert len(pids) == 2
    time.sleep(0.5)
    for pid in pids:
        os.kill(pid, 0)                          # alive
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=40)
    assert rc == 128 + signal.SIGTERM
    t_end = time.monotonic() + 15
    alive = pids
    while alive and time.monotonic() < t_end:
        nxt = []
        for pid in alive:
            try:
                os.kill(pid, 0)
                nxt.append(pid)
            except ProcessLookupError:
                pass
        alive = nxt
        time.sleep(0.1)
    assert not alive, f"ranks {alive} outlived their launcher"
    assert "stopping the ranks" in p.stderr.read()
