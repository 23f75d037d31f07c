#!/usr/bin/env python3
"""Headline benchmark: walker-steps/s of the emcee stretch-move hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): 65 536 walkers x 32-dim isotropic Gaussian, fp64,
a = 2, 10^4 generations (burn-in = first half), streaming moments ON, chain storage OFF.
One bench "step" = GENS_PER_STEP (1000) generations = 65.536e6 walker-steps per GPU, so the
default --steps 10 is exactly the 10^4-generation job.  With --gpus N (launched by
torch.distributed.run, one rank per GPU) the ensemble is 65 536 x N walkers, walker-sharded with the reference's EXACT
partner rule and a peer-to-peer exchange over xGMI (KMC_P2P, pull of the drawn partner rows with system-scope loads +
signal kernel; admitted by a bit-exact self-check, and the timed run itself is verified against the unsharded run;
the push of accepted rows is the second rung, the faster admitted one supplies `value`;
KMC_BENCH_EXCHANGE=pull keeps the pull, =allgather -- or any failure -- runs the native RCCL all-gather
of the updated half per half-step, enqueued by the library itself) -- weak scaling, config C4 at N = 8.  Extra key `dealt_mode` (N > 1, never `value`):
the same job as dealt sub-ensembles, one RCCL all_to_all_single per 64 generations instead of an exchange per half-step.

`python3 bench.py --gpus N` from a plain shell (no WORLD_SIZE in the environment) starts its own N ranks -- child processes,
one per GPU, created BEFORE this process touches the GPU -- relays rank 0's JSON line and exits with the job's status; under
`python -m torch.distributed.run` (WORLD_SIZE set) it is one rank of the job, as the driver launches it.

Prints ONE JSON line (rank 0).  `value` = all walker-steps of the timed region / wall time
(max over ranks) with the ensemble resident in HBM.  `roofline` prices the half-step kernel
against the 8 TB/s HBM spec using ALGORITHMIC read bytes ((2*ndim+1)*8 B per walker-step,
SURVEY.md 8(d)); `cpu_baseline` times the CPU oracle (a port of the reference algorithm, not
KissMCMC.jl itself) on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def note(msg: str) -> None:
    """One line on stderr in ONE write: the ranks share the terminal, and print() hands the text and the newline over separately."""
    sys.stderr.write(msg + "\n")
    sys.stderr.flush()


def bench_test_opt(name: str, default=None):
    """KMC_BENCH_TEST="opt[=value],opt,...": the switches only the tests of this file use (README): backend=gloo and walkers=n
    (several ranks rehearsed on ONE GPU), force-sharded (the N > 1 code path with one rank over real RCCL), fault=point:rank (one rank
    fails at a point of the ladder), no-allgather-extra, deal-epoch=n, no-hbm-shapes, timeout=s (the whole N > 1 job, default 1500) and rung-timeout=s
    (each rung of its ladder, default 300: a hung collective ends in a non-zero exit)."""
    for item in os.environ.get("KMC_BENCH_TEST", "").split(","):
        k, _, v = item.partition("=")
        if k == name:
            return v if v else True
    return default


NWALKERS_PER_GPU = int(bench_test_opt("walkers", 65536))   # (override: rehearsing several ranks on ONE GPU only)
NDIM = 32
GENS_PER_STEP = 1000
REPS = 3                # timed repetitions of the whole job after the warm-up: `value` is their median, `value_min` / `value_max` the spread
SEED = 12345
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy rate is 6290 GB/s
HBM_COPY_GBS = 6290.0   # what a float4 copy achieves on this chip (MI355X_MICROARCH.md:36)
MALL_BYTES = 256 << 20  # Infinity Cache: a working set below this is served on-die between two uses (MI355X_MICROARCH.md:303-310)


def kernel_geometry(describe: str):
    """The part of kmc_sampler_describe that names the kernel and its launch geometry, e.g.
    'half_step_vec L=8 K=2 ITER=2 exact-size, grid 1024 x 128' (or 'generation_group L=16 K=2, ..., grid 2048 x 128': one launch per generation) --
    what a tracked profile record is matched on."""
    import re
    m = re.search(r"(?:half_step|generation)_\w+[^;]*?, grid \d+ x \d+", describe or "")
    return m.group(0) if m else None


def launches_per_generation(describe: str) -> int:
    """2: one launch per half-step (src/samplers.jl:246-273); 1: the one-launch-per-generation kernels (kmc_generation.hpp)."""
    return 1 if "one launch per generation" in (describe or "") else 2


def launch_mode_of(describe: str) -> str:
    """How the launches of `describe` were issued: 'updated_graph_128' / 'table_graph_64' / 'eager' / 'single' (kissmcmc_hip.h: KMC_LAUNCH_*) -- a profile
    record is only evidence for a line that ran in the record's mode."""
    import re
    m = re.search(r"hipGraph replay of (\d+) generations( with per-replay parameter updates)?", describe or "")
    if m:
        return ("updated_graph_" if m.group(2) else "table_graph_") + m.group(1)
    return "eager" if "eager launches" in (describe or "") else "single"


def kernel_name(pdf, describe: str):
    """The template instance the describe string stands for, as the kernel trace names it (density first, then L, K, ITER)."""
    import re
    m = re.search(r"(half_step_\w+) L=(\d+) K=(\d+) ITER=(\d+) (ragged|exact-size)", describe or "")
    if m:
        return f"{m.group(1)}<{type(pdf).__name__}, L={m.group(2)}, K={m.group(3)}, ITER={m.group(4)}, {m.group(5)}>"
    m = re.search(r"(generation_group) L=(\d+) K=(\d+)", describe or "")
    if m:
        return f"{m.group(1)}<{type(pdf).__name__}, L={m.group(2)}, K={m.group(3)}>"
    m = re.search(r"(?:half_step|generation)_\w+", describe or "")
    return f"{m.group(0)}<{type(pdf).__name__}>" if m else (describe or "").split(":")[0]


def moment_bytes(describe: str) -> int:
    """Bytes of streaming-moment accumulators the vector kernel of `describe` touches (kmc_kernels.hpp: accumulate_wave): the
    transposed fold (K = 2, L = 8/16/32) keeps 8 L / 64 doubles per thread, the plain one 2 K double2 per thread of group 0's slots."""
    import re
    m = re.search(r"L=(\d+) K=(\d+) ITER=\d+ \S+, grid (\d+) x (\d+)", describe or "") or re.search(r"generation_group L=(\d+) K=(\d+),.*?, grid (\d+) x (\d+)", describe or "")
    if not m:
        return 0
    L, K, grid, tpb = (int(v) for v in m.groups())
    threads = grid * tpb
    if "generation_group" in describe and not (K == 2 and L in (8, 16, 32)):
        return 0                     # (per-walker sums laid out like the rows: counted with the rows' walkers by the caller -- not a geometry the bench runs)
    return threads * (8 * L // 64) * 8 if (K == 2 and L in (8, 16, 32)) else threads * 2 * K * 16


def state_bytes(nrows: int, ndim: int, moments: int = 0) -> int:
    """Bytes a generation touches: the rows (ld = ndim rounded up to even), the per-walker block {logp f64, naccept u32, klast u32}
    and the streaming-moment accumulators."""
    ld = ndim + (ndim & 1)
    return nrows * ld * 8 + nrows * 16 + moments


def kernel_sources_sha16() -> str:
    """sha256[:16] of the three headers the half-step and generation kernels are made of -- what a profile record was taken from."""
    import hashlib
    h = hashlib.sha256()
    for f in ("kmc_kernels.hpp", "kmc_device.hpp", "kmc_generation.hpp"):
        h.update(open(os.path.join(ROOT, "kissmcmc.jl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def profile_record(name: str, geometry):
    """profiles/traffic_<name>.json -- the tracked record of the rocprofv3 PMC passes and the -DKMC_PROBE build for ONE kernel geometry
    (scripts/profile_r04.sh + scripts/summarize_r04.py write it).  It is only used when the run it is attached to executed that
    very geometry; otherwise (None, reason)."""
    path = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
    if not os.path.exists(path):
        return None, f"profiles/traffic_{name}.json is missing"
    try:
        rec = json.load(open(path))
    except Exception as e:  # noqa: BLE001
        return None, f"profiles/traffic_{name}.json is unreadable ({e})"
    rec["kernel_sources_unchanged"] = rec.get("kernel_sources_sha16") == kernel_sources_sha16()     # (reported, not a reason to refuse)
    if rec.get("geometry") != geometry:
        return None, (f"profiles/traffic_{name}.json was taken from '{rec.get('geometry')}', this run executed '{geometry}': refused")
    return rec, None


def roofline_block(pdf, describe: str, nwalkers_launch: int, ndim: int, launch_us: float, launches: int, state_b: int, record_name: str, use_record: bool = True):
    """`roofline` of the dominant kernel (one launch = one half-step; the callers pass walkers and microseconds PER HALF-STEP, and where the sampler runs one launch
    per generation -- `half_steps_per_launch` 2 -- both are doubled here): numbers and short enums only (the prose lives in DESIGN.md section 5).
    achieved / frac = ALGORITHMIC read bytes ((2 ndim + 1) * 8 per walker-step) / average launch-to-launch time from HIP events over
    the timed region (includes the kernel boundary), priced against the 8 TB/s HBM spec (`peak`).  `served_from` says where the rows really
    come from (a state below 256 MiB lives in the Infinity Cache between two generations) and `bound` follows it.  What is not measured in
    this run -- `traffic` (rocprofv3 --pmc passes: 2 x FETCH_SIZE + WRITE_SIZE per launch, gfx950 read correction) and the in-kernel body /
    boundary split (-DKMC_PROBE build; body_frac = the same bytes / body_us / 8 TB/s) -- comes from the tracked record of that very kernel
    geometry, profiles/traffic_<record>.json (scripts/profile_r04.sh), or is null when the geometries differ."""
    b_read, b_total = (2 * ndim + 1) * 8, (3 * ndim + 2) * 8
    hs = 2 if launches_per_generation(describe) == 1 else 1
    nwalkers_launch, launch_us = nwalkers_launch * hs, launch_us * hs
    alg_read = nwalkers_launch * b_read
    achieved = alg_read / (launch_us * 1e-6) / 1e9
    geometry = kernel_geometry(describe)
    rec, why = profile_record(record_name, geometry) if use_record else (None, "not a single-GPU run")
    served = "infinity_cache" if state_b <= MALL_BYTES else "hbm"
    body_us = rec.get("body_us") if rec else None
    boundary_us = rec.get("boundary_us") if rec else None
    body_frac = (alg_read / (body_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if body_us else None
    # the kernel-duration figure between the two (`duration_source`): "rocprof_trace" -- the trace's average duration where the tool does not inflate it --, else
    # "light_probe_stamps" -- the record says `rocprof_inflated` -- first wave in .. last store issued of the light in-kernel probe (then equal to body_frac)
    duration_us = rec.get("duration_us") if rec else None
    duration_frac = (alg_read / (duration_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if duration_us else None
    if served == "hbm":
        limited = "hbm_bandwidth"
    elif boundary_us and boundary_us / launch_us >= 0.2:
        limited = "kernel_boundary+cache_latency"
    else:
        limited = "cache_latency"
    return {"bound": served, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": rec.get("hbm_bytes_per_launch") if rec else None,
            "priced_against": "hbm_spec", "frac_of_measured_copy_rate": achieved / HBM_COPY_GBS,
            "kernel": kernel_name(pdf, describe), "geometry": geometry, "launches": launches, "avg_launch_us": launch_us, "half_steps_per_launch": hs,
            "algorithmic_read_bytes_per_launch": alg_read, "algorithmic_total_bytes_per_launch": nwalkers_launch * b_total,
            "state_bytes": state_b, "served_from": served, "limited_by": limited,
            "body_us": body_us, "boundary_us": boundary_us, "body_frac": body_frac,
            "duration_us": duration_us, "duration_frac": duration_frac, "duration_source": rec.get("duration_source") if rec else None,
            "launch_mode": launch_mode_of(describe),
            "profile_record": ({"record": f"profiles/traffic_{record_name}.json",
                                **{k: rec.get(k) for k in ("head", "kernel_sources_unchanged", "launch_mode", "period_us_unprofiled", "rocprof_avg_duration_us", "rocprof_inflated",
                                                           "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "l2_hit_rate", "source")}}
                               if rec else {"refused": why})}


def theta0_c2(nwalkers: int) -> np.ndarray:
    """theta0 = 0 + 0.1 * N(0, I): make_theta0s(zeros(32), 0.1, pdf, nwalkers) (reference test default ball_radius)."""
    import kissmcmc_jl_amd as kmc
    return kmc.make_theta0s(np.zeros(NDIM), 0.1, kmc.GaussianIso(), nwalkers, rng=SEED)


def host_threads() -> int:
    """Threads this process may actually run on: the affinity mask, capped by the cgroup's CPU quota (a GPU box's
    container sees all 256 hardware threads of the host in its mask but is throttled to its share, 16 CPUs for one GPU:
    256 OpenMP threads on that share ran 100x slower than 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                            # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except Exception:  # noqa: BLE001
        try:                                        # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:  # noqa: BLE001
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def cpu_baseline(budget_s: float = 10.0):
    """The CPU leg, in a process of its own: OpenMP reads its thread placement when it starts, and this process has long
    started it (torch).  The worker runs with the threads BOUND to adjacent cores (OMP_PROC_BIND=close, OMP_PLACES=cores unless the
    caller set them): the 16 MB ensemble then stays in one or two CCDs' L3 instead of following threads that float over both
    sockets of the host -- measured on the GPU box, 16 threads: 0.7-1.35e8 walker-steps/s unbound (run to run), 1.5-1.65e8 bound."""
    import subprocess
    env = os.environ.copy()
    env.setdefault("OMP_PROC_BIND", "close")
    env.setdefault("OMP_PLACES", "cores")
    env.pop("OMP_NUM_THREADS", None)       # (a launcher's per-rank bound -- torch.distributed.run sets 1 -- is not this leg's: it asks for its threads itself)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(budget_s)], env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"value"')]
    if r.returncode != 0 or not lines:
        return {"error": f"the CPU baseline worker failed (status {r.returncode}): {r.stderr[-500:]}"}
    return json.loads(lines[-1])


def cpu_baseline_worker(budget_s: float = 10.0):
    """Oracle (C + OpenMP over the active half, like Threads.@threads at src/samplers.jl:248; streaming moments summed
    in per-thread blocks) on C2's shape -- same inputs, seed, burn-in and moment settings as the GPU run -- for a bounded
    number of generations, on all the threads this process may use and on one."""
    import oracle
    oracle.build()
    cores = host_threads()
    th = theta0_c2(NWALKERS_PER_GPU)

    def run(G, nthreads):
        cfg = oracle.make_config(oracle.GAUSSIAN_ISO, [0.0, 1.0], NWALKERS_PER_GPU, NDIM, G, G // 2, 1, 2.0, SEED,
                                 nthreads=nthreads)
        t0 = time.perf_counter()
        r = oracle.emcee(cfg, th, store_chain=False)
        assert r["status"] == 0
        return time.perf_counter() - t0

    def timed(nthreads, budget, repeats=1):
        run(2, nthreads)             # thread pool start, first touch of the state
        run(4, nthreads)             # (the first calls are mostly start-up: 0.2-0.6 s for a few generations, then 2 ms per generation)
        probe, t = 16, run(16, nthreads)
        while t < 0.5 and probe < 20000:                        # grow the probe until it takes half a second
            probe = int(min(20000, max(2 * probe, probe * 0.6 / max(t, 1e-3))))
            t = run(probe, nthreads)
        G = int(max(50, min(20000, budget / repeats / max(t / probe, 1e-6))))
        t = min(run(G, nthreads) for _ in range(repeats))      # (a shared host: the quieter of the runs)
        return NWALKERS_PER_GPU * G / t, G, t

    v_all, g_all, t_all = timed(cores, budget_s, repeats=2)
    v_one, g_one, t_one = (v_all, g_all, t_all) if cores == 1 else timed(1, min(budget_s, 6.0))
    return {"value": v_all, "unit": "walker-steps/s", "cores": cores, "kind": "port",
            "sample_short": f"C2 shape, {g_all} generations = {NWALKERS_PER_GPU * g_all:.3g} walker-steps in {t_all:.1f} s on {cores} threads; 1 thread: {g_one} in {t_one:.1f} s",
            "single_thread_value": v_one, "thread_scaling": v_all / v_one,
            "sample": f"C2 shape (65536 walkers x 32-dim Gaussian, fp64, moments on after burn-in), {g_all} generations = "
                      f"{NWALKERS_PER_GPU * g_all:.3g} walker-steps in {t_all:.1f} s on {cores} threads (affinity mask capped by the cgroup CPU quota; the faster of two such runs); "
                      f"1 thread: {g_one} generations in {t_one:.1f} s",
            "thread_placement": f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}",
            "note": "CPU restatement of the reference algorithm (allocation-free C + OpenMP), not KissMCMC.jl itself (no julia in this image)"}


def rendezvous_port() -> int:
    """A TCP port for a rendezvous on 127.0.0.1, from BELOW the kernel's ephemeral range (32768+): a port the kernel hands out for
    bind(0) can be taken again -- by some process's outgoing connection -- between closing the probe socket and the store's listen()
    (seen once: EADDRINUSE in a 2-rank test).  Ports here are only ever taken by explicit binds; each candidate is checked by binding it."""
    import os
    import random
    import socket
    rnd = random.Random(os.getpid() * 1000003 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rnd.randrange(15000, 30000)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            try:
                sk.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # (last resort: the kernel's choice)
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rank_env_defaults(env, world: int) -> None:
    """What every rank of an N-rank job needs in its environment BEFORE it imports torch -- set by spawn_ranks for the ranks it
    starts and by main() for a rank started by torch.distributed.run (the driver's form):
    HSA_ENABLE_IPC_MODE_LEGACY=0 -- the host driver only supports dmabuf IPC (RCCL, shared device memory);
    OMP_NUM_THREADS -- a GPU box's container sees every hardware thread of the host but is throttled to its share; N ranks x 256
    OpenMP threads on 16 CPUs made the host-driven rungs 100x slower (torch.distributed.run itself sets 1 when it is unset)."""
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // max(1, world))))


def spawn_ranks(n: int, argv) -> int:
    """`--gpus n` without a launcher: start the n ranks ourselves (one child process per rank with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in its environment -- what torch.distributed.run would set), BEFORE anything in this process
    touches the GPU; never exec.  Rank 0's stdout is relayed (its JSON line last), the other ranks' goes to stderr.  The first
    rank that fails takes the others down; the whole job is bounded by KMC_BENCH_TEST=timeout=<seconds> (default 1500) (a hung collective must end
    in a non-zero exit, not in the caller's lease).  Returns the job's exit status."""
    import signal
    import socket
    import subprocess
    port = rendezvous_port()
    procs = []

    def die_with_parent():
        """In the child, before exec: SIGTERM when the launcher dies (even by SIGKILL, which it cannot relay itself)."""
        try:
            import ctypes
            ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG
        except Exception:  # noqa: BLE001
            pass

    # The ranks are sessions of their own (so that exactly their process groups can be signalled): a launcher that is
    # itself told to stop -- the harness's timeout, Ctrl-C -- must take them with it, or they stay on the GPUs as orphans.
    stopped_by = []

    def on_signal(signum, _frame):
        stopped_by.append(signum)
        raise KeyboardInterrupt

    previous = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    try:
        for r in range(n):
            env = os.environ.copy()
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "KMC_BENCH_SELF_SPAWNED": "1"})
            rank_env_defaults(env, n)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, start_new_session=True, preexec_fn=die_with_parent,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=r == 0))
    except KeyboardInterrupt:
        pass
    if not procs:
        return 128 + (stopped_by[-1] if stopped_by else signal.SIGINT)
    note(f"[bench launcher] started ranks: {' '.join(str(p.pid) for p in procs)}")

    def stop_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)        # each rank is its own session: the exact process groups we started
                except ProcessLookupError:
                    pass

    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(bench_test_opt("timeout", 1500))
    status = 0

    def end_ranks(grace: float = 10.0):
        """SIGTERM to every rank's process group, a grace period, then SIGKILL."""
        stop_all(signal.SIGTERM)
        t_end = time.monotonic() + grace
        while any(p.poll() is None for p in procs) and time.monotonic() < t_end:
            time.sleep(0.1)
        stop_all(signal.SIGKILL)

    try:
        if stopped_by:
            raise KeyboardInterrupt
        while any(p.poll() is None for p in procs):
            bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
            if bad or time.monotonic() > deadline:
                status = bad[0] if bad else 124
                note(f"[bench launcher] {'a rank exited with status ' + str(status) if bad else 'the job ran out of time (KMC_BENCH_TEST=timeout=s)'}: stopping the others")
                time.sleep(5.0 if bad else 0.0)      # (a failing rank's peers usually follow by themselves)
                end_ranks()
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        sg = stopped_by[-1] if stopped_by else signal.SIGINT
        note(f"[bench launcher] stopped by signal {sg}: stopping the ranks")
        status = 128 + sg
    finally:
        for sg, h in previous.items():
            signal.signal(sg, signal.SIG_IGN)        # (nothing may interrupt the clean-up itself)
        if any(p.poll() is None for p in procs):
            end_ranks()
        for sg, h in previous.items():
            signal.signal(sg, h)
    for p in procs:
        p.wait()
    reader.join(timeout=5.0)
    if status == 0:
        status = next((p.returncode for p in procs if p.returncode != 0), 0)
    js = [l for l in lines if l.lstrip().startswith('{"metric"')]
    detail = [l for l in lines if l.lstrip().startswith('{"bench_detail"')]
    for l in lines:
        if l not in js[-1:] and l not in detail[-1:]:
            sys.stderr.write(l)
    if detail:
        sys.stdout.write(detail[-1] if detail[-1].endswith("\n") else detail[-1] + "\n")
    if js:
        sys.stdout.write(js[-1] if js[-1].endswith("\n") else js[-1] + "\n")
        sys.stdout.flush()
    elif status == 0:
        note("[bench launcher] rank 0 printed no result line")
        status = 1
    return status if status >= 0 else 128 - status


LADDER = []      # [{"rung", "ok", "s"}]: every rung of the N > 1 ladder this rank went through, in order (printed with the line).  Rung names (short:
#                  the line has 4 KB): rendezvous; p2p-check:<variant> (set-up + bit-exact self-check), p2p-time:<variant> (its short timing), link-probe (one
#                  fabric link with the pull's access pattern + this GPU alone),
#                  p2p-run (warm-up + timed run); allgather-setup (ncclCommInitRank), allgather-run, torch-allgather-run (the same exchange as a
#                  torch collective per half-step); the extras after `value`: dealt-extra, allgather-extra-setup, allgather-extra


PENDING = {"line": None}      # once `value` is measured: a callable(reason) that PRINTS the result as it stands (an extra that hangs must not take it down)


class rung:
    """Bound one rung of the N > 1 ladder (set-up + self-check, a timed run, an extra): when it has not finished after
    `seconds`, this rank reports where it hung and exits non-zero -- the launcher (spawn_ranks, or torch.distributed.run) then
    ends the job.  A collective that never returns cannot be interrupted from Python, hence a watchdog thread + os._exit.
    Every rung leaves a record {rung, ok, s} in LADDER (ok: no exception left the block and nobody cleared `.ok`), so that a first
    hardware run can be diagnosed from the JSON line alone.
    `fatal=False` (the extras that run AFTER `value` has been measured): on expiry rank 0 prints the result line as it stands, the
    extra marked as timed out, and every rank leaves with status 0 -- a hung extra costs its own numbers, not the job's."""

    def __init__(self, what: str, seconds: float = None, fatal: bool = True):
        self.what = what
        self.seconds = float(bench_test_opt("rung-timeout", 300)) if seconds is None else seconds
        self.timer = None
        self.ok = True
        self.fatal = fatal

    def __enter__(self):
        import threading

        def expired():
            rank = os.environ.get("RANK", "0")
            note(f"[rank {rank}] bench.py: '{self.what}' did not finish within {self.seconds:.0f} s (hung collective or "
                 f"peer wait?): giving up; Python stacks of this rank:")
            LADDER.append({"rung": self.what, "ok": False, "s": round(time.perf_counter() - self.t0, 3), "timed_out": True})
            note(f"[rank {rank}] ladder so far: {json.dumps(LADDER)}")
            try:
                import faulthandler
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            except Exception:  # noqa: BLE001
                pass
            if not self.fatal and PENDING["line"] is not None:
                try:
                    if rank == "0":
                        PENDING["line"](f"'{self.what}' did not finish within {self.seconds:.0f} s")
                    sys.stderr.flush()
                    os._exit(0)
                except Exception as e:  # noqa: BLE001
                    note(f"[rank {rank}] could not print the result line ({e})")
            os._exit(3)
        self.timer = threading.Timer(self.seconds, expired)
        self.timer.daemon = True
        self.timer.start()
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        LADDER.append({"rung": self.what, "ok": bool(self.ok and exc[0] is None), "s": round(time.perf_counter() - self.t0, 3)})
        return False


def other_configs(kmc, device: int):
    """'C3, C5: report absolute walker-steps/s and roofline fraction' (+ C1, the README call) --
    driver-timed here, NOT `value`.  Each: the whole job resident in HBM, a warm-up piece, then the timed run (HIP events
    on the sampler's stream), streaming moments on, chain off (C1: chain on, as the README call returns it)."""
    out = {}
    rng = np.random.default_rng(SEED)
    cases = [
        ("C1", "README call: 100 walkers x 1-D exponential, niter=10^5 (1000 generations, 500 burn-in)",
         kmc.Exponential(), 0.5 + 0.1 * np.abs(rng.standard_normal((100, 1))), 1000, dict(store_chain=True, store_logp=True)),
        ("C3", "16384 walkers x 64-dim chained Rosenbrock/20, 10^4 generations (burn-in 5000)",
         kmc.Rosenbrock(), 0.1 * rng.standard_normal((16384, 64)), 10000, dict(moments=True)),
        ("C5", "8192 walkers x 1024-dim isotropic Gaussian, 2000 generations (burn-in 1000)",
         kmc.GaussianIso(), rng.standard_normal((8192, 1024)), 2000, dict(moments=True)),
    ]
    for name, what, pdf, th, G, kw in cases:
        try:
            nw, nd = th.shape
            with kmc.Sampler(pdf, nw, nd, G, G // 2, 1, 2.0, SEED, device=device, **kw) as s:
                s.set_positions(th)
                s.run(min(G, 256))
                s.sync()
                runs, phases = [], []
                for _ in range(2):                   # the whole job twice, the faster kept (the graph modes are fed by the host: a
                    s.set_positions(th)              #  descheduled host thread shows as a slow run -- both are reported)
                    s.run(G // 2)                    # burn-in (no moments credited) ...
                    s.sync()
                    ms_burn = s.last_run_ms()
                    s.run(G - G // 2)                # ... and the stored / credited half, timed apart: the profile records (profiles/traffic_*.json)
                    s.sync()                         #     are taken from launches that credit moments
                    ms_after = s.last_run_ms()
                    runs.append(ms_burn + ms_after)
                    phases.append((ms_burn * 1e3 / (2 * (G // 2)), ms_after * 1e3 / (2 * (G - G // 2))))
                    launches = s.launch_count        # (counted from set_positions)
                ms = min(runs)
                us_half = ms * 1e3 / (2 * G)
                us_burn, us_after = phases[runs.index(ms)]
                how = s.describe()
                roof = roofline_block(pdf, how, nw // 2, nd, us_half, launches, state_bytes(nw, nd, moment_bytes(how) if kw.get("moments") else 0), name.lower())
                rec = {"workload": what, "value": nw * G / (ms * 1e-3), "unit": "walker-steps/s", "us_per_half_step": us_half,
                       "us_per_half_step_runs": [r * 1e3 / (2 * G) for r in runs],
                       "us_per_half_step_burnin": us_burn, "us_per_half_step_after_burnin": us_after,
                       "kernel_launches": launches, "algorithmic_read_GBs": roof["achieved"], "frac_of_8TBs": roof["frac"],
                       "frac_of_measured_copy_rate": roof["frac_of_measured_copy_rate"], "state_bytes": roof["state_bytes"], "served_from": roof["served_from"],
                       "accept_ratio_mean": float(s.accept_ratio().mean()), "execution": how}
                if name != "C1":
                    rec["roofline"] = roof
                if launches_per_generation(how) == 1 and name == "C3":
                    # one launch per generation (8 MiB of state, <= 16 384 walkers): the same job on the two-launch kernels beside it
                    old = os.environ.get("KMC_DEBUG")
                    os.environ["KMC_DEBUG"] = (old + "," if old else "") + "fused=0"
                    try:
                        with kmc.Sampler(pdf, nw, nd, G, G // 2, 1, 2.0, SEED, device=device, **kw) as s2:
                            s2.set_positions(th)
                            s2.run(min(G, 1024))
                            s2.sync()
                            s2.set_positions(th)
                            s2.run(G)
                            s2.sync()
                            rec["two_launches_us_per_half_step"] = s2.last_run_ms() * 1e3 / (2 * G)
                            rec["two_launches_execution"] = s2.describe()
                    finally:
                        if old is None:
                            os.environ.pop("KMC_DEBUG", None)
                        else:
                            os.environ["KMC_DEBUG"] = old
                if kw.get("moments"):
                    msum, msq, n = s.moments()
                    mean = msum / max(1, n)
                    rec["posterior_mean_minmax"] = [float(mean.min()), float(mean.max())]
                out[name] = rec
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)}
    # A working set that really lives in HBM (every configuration BASELINE names is Infinity-Cache resident: 8-64 MiB against 256 MiB):
    # the same kernels, exact rule, on ensembles whose state is 512 MiB -- the initial ensemble drawn on the device (kmc_sampler_init_ball,
    # N(0, I): nothing of that size crosses the link), a warm-up piece, then the timed piece twice.
    for name, nw, nd, G in (("HBM_2Mx32", 2097152, 32, 200), ("HBM_512Kx128", 524288, 128, 200)):
        if bench_test_opt("no-hbm-shapes"):
            break
        try:
            pdf = kmc.GaussianIso()
            with kmc.Sampler(pdf, nw, nd, REPS * G + 64, 64, 1, 2.0, SEED, device=device, moments=True) as s:      # (burn-in = the warm-up piece)
                s.init_ball(np.zeros(nd), np.ones(nd), seed=SEED)
                s.run(64)
                s.sync()
                runs = []
                for _ in range(REPS):
                    l0 = s.launch_count
                    s.run(G)
                    s.sync()
                    runs.append(s.last_run_ms())
                    launches = s.launch_count - l0
                ms = sorted(runs)[len(runs) // 2]          # the median (these launches are two-valued by process, not by run: profiles/r05_hbm_bimodal.txt)
                us_half = ms * 1e3 / (2 * G)
                how = s.describe()
                roof = roofline_block(pdf, how, nw // 2, nd, us_half, launches, state_bytes(nw, nd, moment_bytes(how)), name.lower())
                msum, msq, n = s.moments()
                out[name] = {"workload": f"{nw} walkers x {nd}-dim isotropic Gaussian (state {roof['state_bytes'] / 2**20:.0f} MiB > the 256 MiB Infinity Cache), exact partner rule, "
                                         f"moments on, {G} generations timed {REPS} times after 64 of warm-up (the run continues: no restart; median)",
                             "value": nw * G / (ms * 1e-3), "unit": "walker-steps/s", "us_per_half_step": us_half,
                             "us_per_half_step_runs": [r * 1e3 / (2 * G) for r in runs], "kernel_launches": launches,
                             "algorithmic_read_GBs": roof["achieved"], "frac_of_8TBs": roof["frac"], "frac_of_measured_copy_rate": roof["frac_of_measured_copy_rate"],
                             "accept_ratio_mean": float(s.accept_ratio().mean()), "nmoment": int(n), "execution": how, "roofline": roof}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)}
    # Mid-size ensembles with short rows (where the reference's users live: 10^2-10^4 walkers of a few parameters): one launch per
    # generation (kmc_generation.hpp), and the same job on the two-launch kernels (KMC_DEBUG=fused=0) beside it.
    for name, nw, nd, G in (("MID_4096x4", 4096, 4, 20000), ("MID_16384x4", 16384, 4, 20000)):
        try:
            th = np.random.default_rng(SEED).standard_normal((nw, nd))
            rec = {"workload": f"{nw} walkers x {nd}-dim isotropic Gaussian, {G} generations (burn-in {G // 2}), moments on"}
            for label, dbg in (("", None), ("two_launches_", "fused=0")):
                old = os.environ.get("KMC_DEBUG")
                if dbg:
                    os.environ["KMC_DEBUG"] = (old + "," if old else "") + dbg
                try:
                    with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, 1, 2.0, SEED, device=device, moments=True) as s:
                        s.set_positions(th)
                        s.run(256)
                        s.sync()
                        s.set_positions(th)
                        s.run(G)
                        s.sync()
                        ms = s.last_run_ms()
                        rec[label + "value"] = nw * G / (ms * 1e-3)
                        rec[label + "us_per_half_step"] = ms * 1e3 / (2 * G)
                        rec[label + "kernel_launches"] = s.launch_count
                        rec[label + "execution"] = s.describe()
                        if not dbg:
                            rec["unit"] = "walker-steps/s"
                            rec["accept_ratio_mean"] = float(s.accept_ratio().mean())
                finally:
                    if dbg:
                        if old is None:
                            os.environ.pop("KMC_DEBUG", None)
                        else:
                            os.environ["KMC_DEBUG"] = old
            out[name] = rec
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)}
    # "report also one run with nthin such that the chain fits (e.g. 50 stored samples/walker)" -- the C2 job with its
    # chain and log-pdfs stored (nthin = 100), then read out in the reference's order thetas[w][k] (device transposition + D2H)
    try:
        import time
        nw, nd, G, nthin = 65536, 32, 10000, 100
        th = np.random.default_rng(SEED).standard_normal((nw, nd))
        with kmc.Sampler(kmc.GaussianIso(), nw, nd, G, G // 2, nthin, 2.0, SEED, device=device, moments=True, store_chain=True, store_logp=True) as s:
            s.set_positions(th)
            s.run(256)
            s.sync()
            s.set_positions(th)
            s.run(G)
            s.sync()
            ms = s.last_run_ms()
            t0 = time.perf_counter()
            ch, lp = s.chain(by_walker=True)
            t_read = time.perf_counter() - t0
            us_half = ms * 1e3 / (2 * G)
            out["C2_chain_on"] = {"workload": "C2 with the chain and log-pdfs stored, nthin = 100 (50 samples per walker)", "value": nw * G / (ms * 1e-3),
                                  "unit": "walker-steps/s", "us_per_half_step": us_half, "chain_bytes": int(ch.nbytes + lp.nbytes),
                                  "readout_by_walker_ms": t_read * 1e3, "readout": "thetas[w][k] order: transposed on the device, then one D2H copy per piece (pageable host array)",
                                  "chain_shape": list(ch.shape)}
            del ch, lp
    except Exception as e:  # noqa: BLE001
        out["C2_chain_on"] = {"error": str(e)}
    # the general route for a caller's own log-density: a function body compiled at run time (CDensity), at the C2 shape and at
    # the reference's own.  C2_user_density: the Gaussian as anybody would write it -- a sum over elements, which the library
    # recognises and runs lane-striped like a menu density; C2_user_density_two_sums: a rank-one coupling, -0.5 (sum x_i^2 + c (sum x_i)^2):
    # two sums fed by one pass, recognised as well; C2_user_density_coupled: a body no per-element form can express (second-neighbour
    # coupling, two loops), whose rows travel lane-striped while the body is evaluated once per walker on the whole proposal.
    body = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;"
    two_sums = "double s = 0, t = 0; for (int i = 0; i < n; ++i) { s += x[i] * x[i]; t += x[i]; } return -0.5 * (s + p[0] * t * t);"
    coupled = "double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; for (int i = 0; i + 2 < n; ++i) s += p[0] * x[i] * x[i + 2]; return -0.5 * s;"
    for key, src, params, nw, nd, G in (("C2_user_density", body, [], 65536, 32, 4096), ("C2_user_density_two_sums", two_sums, [0.05], 65536, 32, 4096),
                                        ("C2_user_density_coupled", coupled, [0.2], 65536, 32, 4096), ("C1_user_density", body, [], 100, 1, 20000)):     # (a job planned >= 4096 generations measures its launch modes in the warm-up piece)
        try:
            pdf = kmc.CDensity(src, params=params)
            with kmc.Sampler(pdf, nw, nd, G, G // 2, 1, 2.0, 12345, moments=True) as s:
                s.set_positions(np.random.default_rng(12345).standard_normal((nw, nd)))
                s.run(min(G, 1024))                  # warm-up: code objects, graph instantiation and the launch-mode measurement (>= 896 generations)
                s.sync()
                s.set_positions(np.random.default_rng(12345).standard_normal((nw, nd)))
                s.run(G)
                s.sync()
                ms = s.last_run_ms()
                out[key] = {"workload": f"{nw} walkers x {nd}-D, log-density written as a C function body (hiprtc): {src}", "generations": G, "value": nw * G / (ms * 1e-3),
                            "unit": "walker-steps/s", "us_per_half_step": ms * 1e3 / (2 * G), "recognised_as_sum_over_elements": pdf.separable,
                            "accept_ratio_mean": float(s.accept_ratio().mean()), "execution": s.describe()}
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": str(e)}
    return out




# ---------------------------------------------------------------------------------------------------------------------------------
# The result: a FULL record (every block as measured, with its strings; `bench_detail.json` beside this file and an earlier stdout line
# {"bench_detail": ...}) and the ONE result line, the last line of stdout, which must survive a reader that keeps a few KB of the tail:
# numbers and short enums only, under LINE_LIMIT characters in the N = 1 and the N > 1 form (tests/test_bench_cpu.py).
# ---------------------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4000


def sig(v, digits: int = 6):
    """Floats to `digits` significant digits (the line is read by people and by a 4 KB window); everything else as it is."""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}") if v == v and abs(v) != float("inf") else None
    if isinstance(v, (list, tuple)):
        return [sig(x, digits) for x in v]
    if isinstance(v, dict):
        return {k: sig(x, digits) for k, x in v.items()}
    return v


def clip(text, n: int):
    return text if text is None or len(text) <= n else text[: n - 1] + "~"


def compact_roofline(roof: dict) -> dict:
    keep = ("bound", "achieved", "peak", "unit", "frac", "duration_frac", "body_frac", "duration_source", "launch_mode", "traffic", "served_from", "limited_by", "avg_launch_us", "launches",
            "kernel", "geometry", "algorithmic_read_bytes_per_launch", "state_bytes", "frac_of_measured_copy_rate")
    out = {k: roof.get(k) for k in keep}
    rec = roof.get("profile_record") or {}
    out["profile_record"] = ({"record": rec.get("record"), "head": rec.get("head"), "launch_mode": rec.get("launch_mode"), "rocprof_inflated": rec.get("rocprof_inflated")}
                             if "refused" not in rec else {"refused": True})
    return sig(out)


def compact_line(full: dict) -> str:
    """The result line from the full record: the contract's keys whole, every block reduced to its numbers and enums, other_configs to
    {value, us_per_half_step, frac, served_from}; if it still exceeds LINE_LIMIT the optional blocks go, least important first."""
    cfg = dict(full["config"])
    cfg["execution"] = clip(cfg.get("execution"), 150)
    cfg["parallelism"] = clip(cfg.get("parallelism"), 150)
    cfg["workload"] = clip(cfg.get("workload"), 160)
    line = {k: full[k] for k in ("metric", "value", "value_min", "value_max", "repetitions", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = cfg
    line["roofline"] = compact_roofline(full["roofline"])
    cb = full.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = ({"error": clip(str(cb["error"]), 160)} if "error" in cb else
                                sig({**{k: cb.get(k) for k in ("value", "unit", "cores", "kind", "single_thread_value")}, "sample": clip(cb.get("sample_short") or cb.get("sample"), 140)}))
    line["check"] = sig(full["check"], 8)
    if "collective" in full:
        c = full["collective"]
        line["collective"] = {**{k: c.get(k) for k in ("backend", "world_size", "ranks_seen_by_all_reduce", "rccl_version", "native_rccl_version", "rank_env")},
                              "launcher": clip(c.get("launcher"), 16)}
        line["value_from"] = full.get("value_from")
        line["ladder"] = [{k: r[k] for k in ("rung", "ok", "s", "timed_out") if k in r} for r in full.get("ladder", [])]
        for key, keep in (("dealt_mode", ("value", "epoch_generations", "deals", "accept_ratio_mean", "posterior_mean_absmax", "posterior_var_minmax")),
                          ("allgather_mode", ("value", "generations", "us_per_half_step", "equals_unsharded_run"))):
            if key in full:
                b = full[key]
                line[key] = {"error": clip(str(b["error"]), 120)} if "error" in b else sig({k: b.get(k) for k in keep})
                if key == "allgather_mode" and "error" not in b:
                    line[key]["captured_in_graph"] = "captured in the graph" in (b.get("execution") or "")
        if "extras_timed_out" in full:
            line["extras_timed_out"] = clip(full["extras_timed_out"], 120)
        fab = full.get("fabric") or {}
        line["fabric"] = sig({k: fab.get(k) for k in ("bytes_per_link_per_launch", "link_gather_GBs", "link_copy_GBs", "local_gather_GBs", "link_rate_source", "link_bound_us", "single_gpu_us_per_launch",
                                                      "projected_exact_speedup", "measured_speedup", "ge6x_expected_under_exact_rule", "push_bytes_per_link_per_launch", "variants_us_per_launch")
                              if fab.get(k) not in (None, {})}, 4)
    isl = full.get("island_mode")
    if isl is not None:
        line["island_mode"] = {"error": clip(str(isl["error"]), 80)} if "error" in isl else sig({k: isl.get(k) for k in ("value", "accept_ratio_mean")})
    oc = full.get("other_configs")
    if oc:
        line["other_configs"] = {}
        for name, e in oc.items():
            if "error" in e:
                line["other_configs"][name] = {"error": clip(str(e["error"]), 60)}
                continue
            c = {"value": e.get("value"), "us_per_half_step": e.get("us_per_half_step")}
            if name.startswith("HBM_") and e.get("us_per_half_step_runs"):
                c["us_minmax"] = [min(e["us_per_half_step_runs"]), max(e["us_per_half_step_runs"])]
            roof = e.get("roofline")
            if roof:
                c.update(frac=roof["frac"], served_from=roof["served_from"], traffic=roof.get("traffic"))
            if "two_launches_us_per_half_step" in e:
                c["two_launches_us_per_half_step"] = e["two_launches_us_per_half_step"]
            line["other_configs"][name] = sig(c, 5)
    line["detail"] = "bench_detail.json; earlier stdout line {\"bench_detail\": ...}"

    def size():
        return len(json.dumps(line))

    # too long (many ladder rungs, long error strings): the optional blocks go, least important first
    for drop in (lambda: line.pop("island_mode", None),
                 lambda: line.__setitem__("other_configs", {k: {kk: v.get(kk) for kk in ("value", "frac", "error") if kk in v} for k, v in line.get("other_configs", {}).items()}) if "other_configs" in line else None,
                 lambda: line.pop("fabric", None),
                 lambda: line.__setitem__("ladder", [[r["rung"], r["ok"], r["s"]] for r in line["ladder"]]) if "ladder" in line else None,
                 lambda: line.pop("other_configs", None),
                 lambda: line["config"].__setitem__("execution", clip(line["config"].get("execution"), 60)),
                 lambda: line.__setitem__("ladder", line["ladder"][-8:]) if "ladder" in line else None):
        if size() <= LINE_LIMIT:
            break
        drop()
    text = json.dumps(line)
    assert len(text) <= LINE_LIMIT, f"the result line is {len(text)} characters"
    return text


def emit(full: dict) -> None:
    """Rank 0: the full record (file + an earlier stdout line), then the result line -- the LAST line of stdout."""
    detail = json.dumps({"bench_detail": full})
    try:
        with open(os.path.join(ROOT, "bench_detail.json"), "w") as f:
            f.write(json.dumps(full, indent=1) + "\n")
    except OSError as e:
        note(f"bench.py: could not write bench_detail.json ({e})")
    sys.stdout.write(detail + "\n")
    sys.stdout.write(compact_line(full) + "\n")
    sys.stdout.flush()


class Job:
    """What every leg of a run shares: the arguments, this rank's place in the job, the workload."""

    def __init__(self, args, torch, kmc):
        self.args, self.torch, self.kmc = args, torch, kmc
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        # KMC_BENCH_TEST=force-sharded (testing): take the N > 1 code path with ONE rank -- the whole ladder over the real collective
        # backend (RCCL communicator of one rank, captured all-gathers, all_to_all_single of the dealt mode) on a one-GPU box
        self.sharded = self.world > 1 or bool(bench_test_opt("force-sharded"))
        self.dist = None
        self.collective = None
        self.nw = NWALKERS_PER_GPU * self.world
        self.G = args.steps * GENS_PER_STEP
        self.nburn = self.G // 2
        self.pdf = kmc.GaussianIso()
        self.th = theta0_c2(self.nw)

    def all_ok(self, flag: bool) -> bool:
        """A vote: true only if every rank says so (nobody enters the next collective alone)."""
        t = self.torch.tensor([1.0 if flag else 0.0], device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return t.item() != 0

    def fault(self, point: str):
        """KMC_BENCH_TEST=fault=<point>:<rank> (testing): that rank fails at that point of the ladder -- every rank must then take
        the next rung together."""
        if bench_test_opt("fault") == f"{point}:{self.rank}":
            raise RuntimeError(f"injected fault at {point}")
        if bench_test_opt("fault") == f"{point}_hang:{self.rank}":      # ... or never comes back (the peers then block in their next collective)
            time.sleep(10 ** 6)

    def unsharded(self, gens):
        """(positions, naccept, moments) of the whole ensemble after `gens` generations on ONE GPU (this rank's)."""
        with self.kmc.Sampler(self.pdf, self.nw, NDIM, self.G, self.nburn, 1, 2.0, SEED, moments=True, device=self.local_rank) as ref:
            ref.set_positions(self.th)
            ref.run(gens)
            ref.sync()
            return ref.positions(), ref.naccept(), ref.moments()

    def timed(self, d, gens, warm):
        """`warm` generations, restart, then `gens` generations from a common start: seconds (this rank's; the caller takes the max)."""
        torch, dist = self.torch, self.dist
        if warm > 0:
            d.set_positions(self.th)
            d.run(warm)
            d.sync()
        d.set_positions(self.th)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.run(gens)
        d.sync()
        torch.cuda.synchronize()
        dist.barrier()
        return time.perf_counter() - t0

    @staticmethod
    def take_median(res: dict, runs) -> None:
        """`runs`: (seconds -- max over ranks --, HIP-event ms, launches) per repetition; `value` comes from the median one."""
        res["elapsed_runs"] = [r[0] for r in runs]
        res["elapsed"], res["event_ms"], res["launches"] = sorted(runs)[len(runs) // 2]

    def max_over_ranks(self, seconds: float) -> float:
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def rendezvous(job: Job) -> None:
    """The process group of an N > 1 job (one rank per GPU, "nccl" = RCCL), and what the collective backend really saw."""
    torch = job.torch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    import datetime
    import torch.distributed as dist
    job.dist = dist
    backend = bench_test_opt("backend", "nccl")
    with rung("rendezvous"):
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", job.local_rank), timeout=datetime.timedelta(seconds=600))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=600))
        seen = torch.ones(1, device="cuda")              # did the backend see every rank?  one all-reduce of ones (RCCL over xGMI for "nccl")
        dist.all_reduce(seen)
        torch.cuda.synchronize()
    try:
        rv = torch.cuda.nccl.version()
        rccl_version = ".".join(str(v) for v in rv) if isinstance(rv, tuple) else str(rv)
    except Exception:  # noqa: BLE001
        rccl_version = None
    job.collective = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen_by_all_reduce": int(seen.item()),
                      "rccl_version": rccl_version, "native_rccl_version": job.kmc.Sampler.rccl_version(),
                      "launcher": "bench.py itself (one child process per rank)" if os.environ.get("KMC_BENCH_SELF_SPAWNED") else "external (torch.distributed.run)",
                      "rank_env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS")}}


def run_single(job: Job) -> dict:
    """N = 1: the whole C2 job on one GPU -- `value` -- and, NOT `value`, the opt-in island mode on the same job."""
    kmc, torch, args = job.kmc, job.torch, job.args
    s = kmc.Sampler(job.pdf, job.nw, NDIM, job.G, job.nburn, 1, 2.0, SEED, moments=True, device=job.local_rank)
    s.set_positions(job.th)
    for _ in range(args.warmup):
        s.run(GENS_PER_STEP)
    s.sync()
    runs = []
    for _ in range(REPS):                    # the whole C2 job REPS times from the same start (>= 3 repetitions after a warm-up run)
        s.set_positions(job.th)              # restart: the timed region is the whole C2 job
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run(job.G)                         # exactly `steps` steps of GENS_PER_STEP generations
        s.sync()
        torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0, s.last_run_ms(), s.launch_count))      # (wall; HIP events on the sampler's own stream; launches since set_positions)
    res = {"how": s.describe(), "mode": "single"}
    Job.take_median(res, runs)               # `value`: the median repetition
    # not `value`: the same job once more in its two phases -- burn-in (no moments credited) and the credited half -- because the profile records
    # (profiles/traffic_c2.json) are taken per phase: launches that credit moments are ~0.4 us longer
    s.set_positions(job.th)
    phase = []
    for gens in (job.nburn, job.G - job.nburn):
        l0 = s.launch_count
        s.run(gens)
        s.sync()
        phase.append(s.last_run_ms() * 1e3 / max(1, s.launch_count - l0))
    res["phase_us"] = phase
    res["msum"], res["msq"], res["nmom"] = s.moments()
    res["acc"] = float(s.accept_ratio().mean())
    s.close()
    # Extra, NOT `value`: the opt-in island mode (256-walker islands resident in LDS, partners drawn inside the island, walkers
    # re-dealt every 64 generations) on the same job.
    try:
        if args.no_island:
            raise RuntimeError("skipped (--no-island)")
        with kmc.Sampler(job.pdf, job.nw, NDIM, job.G, job.nburn, 1, 2.0, SEED, moments=True, device=job.local_rank, island_gens=64, island_size=256) as si:
            si.set_positions(job.th)
            si.run(GENS_PER_STEP)
            si.sync()
            si.set_positions(job.th)
            si.run(job.G)
            si.sync()
            ims = si.last_run_ms()
            isum, isq, inm = si.moments()
            imean = isum / max(1, inm)
            res["island"] = {"value": float(job.nw) * job.G / (ims * 1e-3), "unit": "walker-steps/s", "island_size": 256, "island_gens": 64,
                             "accept_ratio_mean": float(si.accept_ratio().mean()), "posterior_mean_absmax": float(np.abs(imean).max()),
                             "posterior_var_minmax": [float((isq / inm - imean ** 2).min()), float((isq / inm - imean ** 2).max())],
                             "note": "KMC_ISLANDS: same target distribution, partner pool = the island's complementary half "
                                     "(not the reference's whole-ensemble rule); bit-exact against the oracle's island restatement"}
    except Exception as e:  # noqa: BLE001
        res["island"] = {"error": str(e)}
    return res


P2P_VARIANTS = [   # (rung tag, what it is, push).  Both read partner rows with system-scope loads; each is admitted by its bit-exact self-check and timed, the
    #                    faster admitted one runs.  KMC_BENCH_EXCHANGE=pull: the first only.
    ("pull", "pull of drawn rows (system-scope loads), signal kernel", False),
    ("push", "push of accepted rows into local copies read with system-scope loads, signal kernel", True),
]


def try_p2p(job: Job, finegrained, push=False):
    """Set up the peer-to-peer exchange and self-check it: 240 generations (hipGraph replays + an eager tail) must reproduce, bit
    for bit, the same generations of the whole ensemble on ONE GPU (rank 0 runs it unsharded).  Any error, time-out or mismatch
    on any rank -> None on every rank."""
    from kissmcmc_jl_amd.distributed import P2PEmcee
    rank = job.rank
    d, ok = None, True
    try:
        job.fault("p2p_setup")
        d = P2PEmcee(job.pdf, job.nw, NDIM, job.G, job.nburn, 1, 2.0, SEED, device=job.local_rank, finegrained=finegrained, push=push, connect=False)      # local part only: no collective yet
    except Exception as e:  # noqa: BLE001
        note(f"[rank {rank}] p2p set-up failed ({e})")
        ok = False
    if job.all_ok(ok):                                   # every rank has its sampler and handles: now the exchange
        try:
            d.connect()
            job.fault("p2p_connect")
        except Exception as e:  # noqa: BLE001
            note(f"[rank {rank}] p2p connect failed ({e})")
            ok = False
    if job.all_ok(ok):
        vgen = 240
        try:
            d.set_positions(job.th)
            d.run(vgen)
            d.sync()                                      # (a peer wait that timed out surfaces here, on the ranks that waited)
            job.fault("p2p_selfcheck")
        except Exception as e:  # noqa: BLE001
            note(f"[rank {rank}] p2p self-check failed ({e})")
            ok = False
        ok = job.all_ok(ok)                               # every rank's kernels ran through: only then the result collectives
        try:
            if not ok:
                raise RuntimeError("a rank failed before the results were gathered")
            vpos, vacc = d.positions(), d.naccept()
            if rank == 0:
                rpos, racc, _ = job.unsharded(vgen)
                if not (np.array_equal(rpos, vpos) and np.array_equal(racc, vacc)):
                    note(f"[rank 0] p2p self-check (finegrained={finegrained}, push={push}): "
                         "sharded run differs from the single-GPU run")
                    ok = False
        except Exception as e:  # noqa: BLE001
            note(f"[rank {rank}] p2p self-check failed ({e})")
            ok = False
        ok = job.all_ok(ok)
    else:
        ok = False
    if not ok:
        if d is not None:
            try:
                d.sampler.close()
            except Exception:  # noqa: BLE001
                pass
        return None
    return d


def make_allgather(job: Job, what, fatal=True):
    """The native all-gather exchange, set up in two votes: every rank's local part (its replica sampler), then the collective part
    (unique id, ncclCommInitRank, the capture vote).  The connected driver, or None on EVERY rank."""
    from kissmcmc_jl_amd.distributed import AllGatherEmcee
    d, ok = None, True
    try:
        job.fault("allgather_setup")
        d = AllGatherEmcee(job.pdf, job.nw, NDIM, job.G, job.nburn, 1, 2.0, SEED, device=job.local_rank, connect=False)
    except Exception as e:  # noqa: BLE001
        note(f"[rank {job.rank}] native RCCL all-gather set-up failed ({e})")
        ok = False
    if job.all_ok(ok):
        try:
            with rung(what, fatal=fatal):
                d.connect()
                job.fault("allgather_connect")
        except Exception as e:  # noqa: BLE001
            note(f"[rank {job.rank}] native RCCL all-gather set-up failed ({e})")
            ok = False
        ok = job.all_ok(ok)
    else:
        ok = False
    if not ok and d is not None:
        try:
            d.sampler.close()
        except Exception:  # noqa: BLE001
            pass
    return d if ok else None


def run_sharded(job: Job) -> dict:
    """N > 1: walker-sharded, one rank per GPU, EXACT partner rule (reference src/samplers.jl:250: partners from the whole complementary
    half).  The ladder: peer-to-peer partner reads over xGMI (KMC_P2P: only the rows that are drawn cross the fabric, the whole run is
    enqueued like the single-GPU case) -- by default ONE variant, pull of the drawn rows with system-scope loads + a signal kernel (the
    one whose correctness does not depend on cache state), then the push of accepted rows (same loads), each admitted by a bit-exact self-check, the
    faster one runs; KMC_BENCH_EXCHANGE=allgather (or any failure) uses the RCCL
    all-gather of the updated half per half-step.  The TIMED run itself is then verified against the unsharded run of the whole
    ensemble on rank 0.  Returns the measurements; `value` stands once this returns (the extras run afterwards)."""
    torch, dist, args, rank, world = job.torch, job.dist, job.args, job.rank, job.world
    G, nw, nburn, th = job.G, job.nw, job.nburn, job.th
    from kissmcmc_jl_amd.distributed import HipShardExecutor, ShardedEmcee
    mode = os.environ.get("KMC_BENCH_EXCHANGE", "p2p")
    res = {"tried": [], "verified": None, "p2p_variant": None}
    drv, best_t = None, None
    if mode in ("p2p", "pull"):
        cands = P2P_VARIANTS[:1] if mode == "pull" else P2P_VARIANTS
        for tag, label, push in cands:
            with rung(f"p2p-check:{tag}") as rg:
                cand = try_p2p(job, False, push)
                rg.ok = cand is not None
            if cand is None:
                continue
            if len(cands) == 1:
                drv, res["p2p_variant"] = cand, label
                break
            with rung(f"p2p-time:{tag}"):
                tc = job.max_over_ranks(job.timed(cand, 1024, 0))      # long enough to reach the steady state of the replayed graphs (16 chunks)
            res["tried"].append((tag, tc))
            if drv is None or tc < best_t:
                if drv is not None:
                    drv.close()
                drv, best_t, res["p2p_variant"] = cand, tc, label
            else:
                cand.close()
        if drv is None:
            with rung("p2p-check:pull-finegrained") as rg:
                drv = try_p2p(job, True)
                rg.ok = drv is not None
            if drv is not None:
                res["p2p_variant"] = "pull of drawn rows, rows in fine-grained memory, signal kernel"
        if rank == 0 and res["tried"]:
            note("[rank 0] p2p variants, s per 1024 generations: " + "; ".join(f"{l}: {t:.4f}" for l, t in res["tried"]))
        if drv is None and rank == 0:
            note("[rank 0] falling back to the RCCL all-gather exchange")
        mode = "p2p" if drv is not None else "allgather"
    if mode == "p2p":
        # one fabric link measured with the pull's own access pattern, and this GPU alone on its 65 536 walkers: what the exact rule can reach (`fabric`)
        with rung("link-probe") as rg:
            lp = drv.link_probe()
            t1 = float("nan")
            try:
                with job.kmc.Sampler(job.pdf, NWALKERS_PER_GPU, NDIM, G, 1024, 1, 2.0, SEED, moments=True, device=job.local_rank) as s1:
                    for gens in (1024, 2048):            # warm-up (launch modes measured), then half burn-in, half credited like the job
                        s1.set_positions(th[:NWALKERS_PER_GPU])
                        s1.run(gens)
                        s1.sync()
                    t1 = s1.last_run_ms() * 1e3 / (2 * 2048)
            except Exception as e:  # noqa: BLE001
                note(f"[rank {rank}] link-probe: the single-GPU reference failed ({e})")
            v = torch.tensor([lp.get("link_gather_GBs", float("nan")), lp.get("link_copy_GBs", float("nan")), lp.get("local_gather_GBs", float("nan")), -t1],
                             dtype=torch.float64, device="cuda")
            dist.all_reduce(v, op=dist.ReduceOp.MIN)     # the slowest link, the slowest GPU
            vals = [None if x != x else x for x in v.tolist()]
            res["link"] = {"link_gather_GBs": vals[0], "link_copy_GBs": vals[1], "local_gather_GBs": vals[2],
                           "single_gpu_us_per_launch": None if vals[3] is None else -vals[3], "rows": lp["rows"], "error": lp.get("error")}
            rg.ok = vals[0] is not None and vals[3] is not None
        ok = True
        try:
            with rung("p2p-run"):
                drv.set_positions(th)
                drv.run(args.warmup * GENS_PER_STEP)
                drv.sync()
                runs = []
                for _ in range(REPS):                    # the whole job REPS times from the same start, as at N = 1
                    drv.set_positions(th)                # barriers inside; restart the job
                    dist.barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    drv.run(G)
                    try:
                        drv.sync()
                        job.fault("p2p_run")
                        ran = True
                    except Exception as e:  # noqa: BLE001  (a timed-out peer wait: the other ranks must not be left inside a collective)
                        note(f"[rank {rank}] the p2p run failed ({e})")
                        ran = False
                    torch.cuda.synchronize()
                    if not job.all_ok(ran):
                        raise RuntimeError("a rank's peer-to-peer run did not complete")
                    dist.barrier()
                    runs.append((job.max_over_ranks(time.perf_counter() - t0), drv.sampler.last_run_ms(), drv.sampler.launch_count))
                job.take_median(res, runs)
                res["msum"], res["msq"], res["nmom"] = drv.moments()
                fpos, facc = drv.positions(), drv.naccept()
                res["how"] = drv.sampler.describe()
                drv.close()
        except Exception as e:  # noqa: BLE001  (e.g. a peer wait that timed out: every rank then takes the fallback)
            note(f"[rank {rank}] the p2p run failed ({e})")
            ok = False
            try:
                drv.sampler.close()
            except Exception:  # noqa: BLE001
                pass
        if not job.all_ok(ok):
            if rank == 0:
                note("[rank 0] falling back to the RCCL all-gather exchange")
            mode = "allgather"
    if mode == "p2p":
        if rank == 0:                                # the timed run itself, against the unsharded run of the whole job on one GPU
            rpos, racc, (rs, rq, rn) = job.unsharded(G)
            res["verified"] = bool(np.array_equal(rpos, fpos) and np.array_equal(racc, facc) and rn == res["nmom"] and
                                   np.allclose(rs, res["msum"], rtol=1e-10, atol=1e-6) and np.allclose(rq, res["msq"], rtol=1e-10, atol=1e-6))
            if not res["verified"]:
                note("[rank 0] the TIMED sharded run differs from the unsharded run of the same job")
        res["parallelism"] = f"walker-sharded x{world}, exact partner rule, peer-to-peer exchange over xGMI (IPC): {res['p2p_variant']}; progress-flag ordering"
        res["value_from"] = "p2p-run"
    else:
        # The exchange the north star names: an RCCL all-gather of the updated half after every half-step.  Native form first
        # (kmc_sampler_run enqueues kernel + ncclAllGather per half-step, inside the hipGraph chunks); if that cannot be set up on
        # every rank, the same exchange as a torch collective per half-step from Python.
        nat = make_allgather(job, "allgather-setup")
        if nat is not None:
            with rung("allgather-run"):
                runs = []
                for rep in range(REPS):
                    t = job.max_over_ranks(job.timed(nat, G, min(args.warmup * GENS_PER_STEP, 200) if rep == 0 else 0))
                    runs.append((t, nat.sampler.last_run_ms(), nat.sampler.launch_count))
                job.take_median(res, runs)
                res["msum"], res["msq"], res["nmom"] = nat.moments()
                facc, fpos = nat.naccept(), nat.positions()
                res["how"] = nat.sampler.describe()
                nat.close()
            res["parallelism"] = f"walker-sharded x{world}, exact partner rule, native RCCL all-gather of the updated half per half-step ({res['how'].split(';')[-1].strip()})"
            res["value_from"] = "allgather-run"
        else:
            with rung("torch-allgather-run"):
                ex = HipShardExecutor(job.pdf, nw, NDIM, G, nburn, 1, 2.0, SEED, rank=rank, world=world, device=job.local_rank)
                ex.set_positions(th)
                sdrv = ShardedEmcee(ex, nw, NDIM)
                sdrv.run(min(args.warmup * GENS_PER_STEP, 100))   # warm-up: kernels + RCCL rings
                ex.sync()
                ex.set_positions(th)
                sdrv.generation = 0
                dist.barrier()
                torch.cuda.synchronize()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                ev0.record()                         # the executor launches on torch's current stream
                sdrv.run(G)
                ev1.record()
                ex.sync()
                torch.cuda.synchronize()
                dist.barrier()
                res["elapsed"] = time.perf_counter() - t0
                res["event_ms"] = ev0.elapsed_time(ev1)
                res["launches"] = 2 * G
                res["msum"], res["msq"], res["nmom"] = sdrv.moments()
                facc, fpos = sdrv.naccept(), sdrv.positions()
                res["how"] = ex.sampler.describe()
                ex.close()
            res["parallelism"] = f"walker-sharded x{world}, exact partner rule, RCCL all-gather of the updated half per half-step (torch collective per half-step)"
            res["value_from"] = "torch-allgather-run"
        if rank == 0:
            rpos, racc, _ = job.unsharded(G)
            res["verified"] = bool(np.array_equal(rpos, fpos) and np.array_equal(racc, facc))
    res["acc"] = float(facc.sum() / nw / max(1, G - nburn))
    if "elapsed_runs" not in res:                        # (the torch-collective fallback: one timed run)
        res["elapsed"] = job.max_over_ranks(res["elapsed"])
        res["elapsed_runs"] = [res["elapsed"]]
    res["mode"] = mode
    res["extras"] = {"dealt": None, "allgather": None, "allgather_started": False}
    return res


def run_extras(job: Job, res: dict) -> None:
    """After `value` (N > 1), never `value`: (1) dealt sub-ensembles -- every GPU runs the reference's algorithm unchanged on its own 65 536
    walkers (partners from its own complementary half) for an epoch, then ONE RCCL all_to_all_single re-deals the walkers across the
    GPUs (state-independent permutation): same target distribution, no per-half-step exchange; (2) when the pull supplied `value`,
    the exchange the north star names -- one RCCL all-gather of the updated half per half-step, enqueued with the kernels inside
    the hipGraph chunks -- on a bounded piece of the same job, so that both exchanges are on record from the same node."""
    torch, dist, args, rank, world = job.torch, job.dist, job.args, job.rank, job.world
    G, nw, nburn = job.G, job.nw, job.nburn
    extras = res["extras"]
    try:
        from kissmcmc_jl_amd.distributed import DealtEmcee, HipDealExecutor
        epoch = int(bench_test_opt("deal-epoch", 64))
        dex, okd = None, True
        try:                                         # local part first, then a vote: nobody enters the collectives alone
            job.fault("dealt_setup")
            dex = HipDealExecutor(job.pdf, NWALKERS_PER_GPU, NDIM, G, nburn, 1, 2.0, SEED, rank=rank, world=world, device=job.local_rank)
        except Exception as e:  # noqa: BLE001
            note(f"[rank {rank}] dealt sub-ensembles: set-up failed ({e})")
            okd = False
        if not job.all_ok(okd):
            if dex is not None:
                dex.close()
            raise RuntimeError("the dealt mode could not be set up on every rank (see stderr)")
        with rung("dealt-extra", fatal=False):
            job.fault("dealt_run")
            dd = DealtEmcee(dex, nw, NDIM, epoch)
            dt = job.max_over_ranks(job.timed(dd, G, args.warmup * GENS_PER_STEP))
            r = dd.results()
            dd.close()
        dmean = r["sum"] / max(1, r["n"])
        dvar = r["sumsq"] / max(1, r["n"]) - dmean ** 2
        extras["dealt"] = {"value": float(nw) * G / dt, "unit": "walker-steps/s", "epoch_generations": epoch, "deals": dd.deals,
                           "all_to_all_bytes_per_gpu_per_deal": NWALKERS_PER_GPU * (NDIM + 2) * 8,
                           "accept_ratio_mean": float(r["naccept"].sum() / nw / max(1, G - nburn)),
                           "posterior_mean_absmax": float(np.abs(dmean).max()),
                           "posterior_var_minmax": [float(dvar.min()), float(dvar.max())], "nmoment": int(r["n"]),
                           "note": "dealt sub-ensembles (kmc_config.deal_count): same target distribution, partner pool = this GPU's complementary "
                                   "half (not the reference's whole-ensemble rule), walkers re-dealt across the GPUs by one RCCL all_to_all_single "
                                   "per epoch; bit-identical to the oracle's restatement kmco_emcee_dealt (tests/test_gpu_dealt.py)"}
    except Exception as e:  # noqa: BLE001
        extras["dealt"] = {"error": str(e)}

    if res["mode"] == "p2p" and not bench_test_opt("no-allgather-extra"):
        try:
            extras["allgather_started"] = True
            ag = make_allgather(job, "allgather-extra-setup", fatal=False)
            if ag is not None:
                with rung("allgather-extra", fatal=False):
                    gens = min(G, 1024)
                    dta = job.max_over_ranks(job.timed(ag, gens, 128))
                    apos, aacc = ag.positions(), ag.naccept()
                    how_ag = ag.sampler.describe()
                    ag.close()
                same = None
                if rank == 0:
                    rp, ra, _ = job.unsharded(gens)
                    same = bool(np.array_equal(rp, apos) and np.array_equal(ra, aacc))
                extras["allgather"] = {"value": float(nw) * gens / dta, "unit": "walker-steps/s", "generations": gens,
                                       "us_per_half_step": dta / (2 * gens) * 1e6, "equals_unsharded_run": same,
                                       "bytes_received_per_gpu_per_half_step": (world - 1) * (NWALKERS_PER_GPU // 2) * NDIM * 8,
                                       "execution": how_ag.split(";")[-1].strip(),
                                       "note": "exact partner rule; full replica per rank, in-place ncclAllGather of the updated half per half-step"}
            else:
                extras["allgather"] = {"error": "native RCCL all-gather could not be set up on every rank (see stderr)"}
        except Exception as e:  # noqa: BLE001
            extras["allgather"] = {"error": str(e)}


def full_record(job: Job, res: dict, timed_out=None) -> dict:
    """Everything measured so far as one record (rank 0; at the end -- or from the watchdog of an extra that hangs)."""
    args, world = job.args, job.world
    value = float(job.nw) * job.G / res["elapsed"]
    mean = res["msum"] / max(1, res["nmom"])
    var = res["msq"] / max(1, res["nmom"]) - mean ** 2
    # dominant kernel: the half-step kernel `how` describes; one launch = one half-step of this rank
    walkers_per_launch = NWALKERS_PER_GPU // 2
    launch_us = res["event_ms"] * 1e3 / max(1, res["launches"])
    rows_here = NWALKERS_PER_GPU if (job.sharded and res["mode"] == "p2p") else job.nw             # (replica modes hold the whole ensemble)
    how = res["how"]
    roof = roofline_block(job.pdf, how, walkers_per_launch, NDIM, launch_us, res["launches"], state_bytes(rows_here, NDIM, moment_bytes(how)), "c2", use_record=not job.sharded)
    if res.get("phase_us"):
        roof["avg_launch_us_burnin"], roof["avg_launch_us_credited"] = res["phase_us"]
    out = {
        "metric": "walker-steps/sec", "value": value, "unit": "walker-steps/s", "n_gpus": world,
        "value_min": float(job.nw) * job.G / max(res["elapsed_runs"]), "value_max": float(job.nw) * job.G / min(res["elapsed_runs"]), "repetitions": len(res["elapsed_runs"]),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["elapsed"] * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C2: emcee stretch move, {NWALKERS_PER_GPU} walkers/GPU x {NDIM}-dim isotropic Gaussian, "
                               f"{job.G} generations (burn-in {job.nburn}), a=2, moments on, chain off",
                   "nwalkers_total": job.nw, "ndim": NDIM, "generations": job.G, "gens_per_step": GENS_PER_STEP,
                   "parallelism": "single GPU" if not job.sharded else res["parallelism"],
                   "execution": how},
        "roofline": roof,
        "check": {"accept_ratio_mean": res["acc"], "posterior_mean_absmax": float(np.abs(mean).max()),
                  "posterior_var_min": float(var.min()), "posterior_var_max": float(var.max()),
                  "nmoment": int(res["nmom"])},
    }
    if job.sharded:
        extras = res["extras"]
        out["collective"] = job.collective
        out["ladder"] = list(LADDER)             # every rung rank 0 went through: {rung, ok, s}
        out["value_from"] = res["value_from"]    # the rung whose timed run is `value`
        out["check"]["timed_run_equals_unsharded_run"] = res["verified"]
        out["dealt_mode"] = extras["dealt"] if extras["dealt"] is not None else {"error": timed_out or "not run"}
        if extras["allgather"] is not None or (timed_out and extras["allgather_started"]):
            out["allgather_mode"] = extras["allgather"] if extras["allgather"] is not None else {"error": timed_out}
        if timed_out:
            out["extras_timed_out"] = timed_out
        # what the exchange has to move: partners are uniform over the whole complementary half (src/samplers.jl:250), so (P-1)/P of a
        # rank's partner rows are remote, 1/P from each peer over that pair's single xGMI link; pull variants move every drawn row once (bytes_per_link from
        # each peer), push variants accepted rows only (push_bytes_per_link to each peer).  The link's rate is MEASURED by the `link-probe` rung with the pull's
        # own access pattern (77 GB/s assumed only if that rung failed: `link_rate_source`); projected_exact_speedup = P t1 / max(t1, link_bound_us), t1 = one
        # GPU alone on 65 536 walkers in this run -- what the reference's partner rule can reach on this fabric, next to measured_speedup.
        rows_per_peer = walkers_per_launch / world
        link = res.get("link") or {}
        rate = link.get("link_gather_GBs")
        t1 = link.get("single_gpu_us_per_launch")
        bound = rows_per_peer * NDIM * 8 / ((rate or 77.0) * 1e9) * 1e6
        proj = world * t1 / max(t1, bound) if t1 else None
        out["fabric"] = {"remote_partner_bytes_per_gpu_per_launch": rows_per_peer * (world - 1) * NDIM * 8,
                         "bytes_per_link_per_launch": rows_per_peer * NDIM * 8,
                         "link_gather_GBs": rate, "link_copy_GBs": link.get("link_copy_GBs"), "local_gather_GBs": link.get("local_gather_GBs"),
                         "link_rate_source": "link-probe" if rate else "assumed_77GBs", "link_bound_us": bound,
                         "single_gpu_us_per_launch": t1, "projected_exact_speedup": proj,
                         "measured_speedup": (value / (NWALKERS_PER_GPU / (2 * t1 * 1e-6))) if t1 else None,
                         "ge6x_expected_under_exact_rule": (proj >= 6.0) if proj else None,      # (north star: >= 6x at 8 GPUs; DESIGN section 7 says no for rand(ncos) over the whole half)
                         "push_bytes_per_link_per_launch": res["acc"] * walkers_per_launch * NDIM * 8,
                         "variants_us_per_launch": {tag: tc / 2048 * 1e6 for tag, tc in res["tried"]}}   # 1024 generations each
    else:
        out["island_mode"] = res.get("island")
    for key in ("other_configs", "cpu_baseline"):
        if key in res:
            out[key] = res[key]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra driver-timed C1/C3/C5 runs (profiling passes)")
    ap.add_argument("--no-island", action="store_true", help="skip the extra island-mode run (profiling passes)")
    ap.add_argument("--test-sleep", type=float, default=0.0, help=argparse.SUPPRESS)   # (tests of the launcher: every rank just sleeps)
    if len(sys.argv) >= 2 and sys.argv[1] == "--cpu-baseline-worker":       # (the CPU leg's own process, see cpu_baseline)
        print(json.dumps(cpu_baseline_worker(float(sys.argv[2]) if len(sys.argv) > 2 else 10.0)), flush=True)
        return
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (nothing in this process has touched the GPU yet, and nothing will)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.test_sleep > 0.0:
        time.sleep(args.test_sleep)
        return

    # a rank of an N-rank job started by somebody else (torch.distributed.run, the driver's form): what spawn_ranks gives the
    # ranks it starts, BEFORE torch -- and with it OpenMP and the HSA runtime -- is loaded
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        rank_env_defaults(os.environ, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"])))

    import torch
    import kissmcmc_jl_amd as kmc

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP emcee path has no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world                            # the launcher decides (python -m torch.distributed.run --nproc-per-node N)
    # one rank per GPU; (testing only: KMC_BENCH_TEST=backend=gloo lets several ranks share one GPU, which RCCL refuses -- the
    # peer-to-peer exchange itself is the same code)
    if world > torch.cuda.device_count() and bench_test_opt("backend", "nccl") == "nccl":
        raise SystemExit(f"bench.py --gpus {world}: one rank per GPU over RCCL, but this node shows {torch.cuda.device_count()} device(s) "
                         "(KMC_BENCH_TEST=backend=gloo rehearses several ranks on fewer devices)")
    job = Job(args, torch, kmc)
    torch.cuda.set_device(job.local_rank)

    if not job.sharded:
        res = run_single(job)
        if not args.no_other_configs:
            res["other_configs"] = other_configs(kmc, job.local_rank)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
    else:
        rendezvous(job)
        res = run_sharded(job)
        # `value` stands from here on: an extra that hangs costs its own numbers only (the watchdog prints the result as it stands)
        PENDING["line"] = lambda reason: emit(full_record(job, res, reason))
        run_extras(job, res)
    if job.rank == 0:
        PENDING["line"] = None
        emit(full_record(job, res))
    if job.dist is not None:
        job.dist.destroy_process_group()


if __name__ == "__main__":
    main()
