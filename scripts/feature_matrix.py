#!/usr/bin/env python3
"""Which sampler feature works with which kind of log-density -- MEASURED, not written down: every cell of the table
creates the sampler through the C ABI (kmc_sampler_create), drives the feature for a few generations and records what
the library answered: "yes", or "no" with the library's own message (KMC_ERR_UNSUPPORTED and friends).

    python3 scripts/feature_matrix.py            # print the table (needs an MI355X)
    python3 scripts/feature_matrix.py --write    # ... and replace the block between the markers in README.md
    python3 scripts/feature_matrix.py --write-from table.md    # paste a table generated elsewhere (the GPU box) into README.md

tests/test_gpu_feature_matrix.py regenerates the table on the GPU box and compares it with README.md, so the two cannot drift.
Rows: the four ways to supply the reference's `pdf` closure (src/samplers.jl:257); columns: execution modes of the hot
loop (src/samplers.jl:245-273) and the state / storage features around it."""
from __future__ import annotations

import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BEGIN, END = "<!-- feature-matrix:begin (scripts/feature_matrix.py --write) -->", "<!-- feature-matrix:end -->"
ND, SEED = 8, 11


def densities(kmc):
    return [
        ("menu density (`GaussianIso`, ...)", lambda: kmc.GaussianIso()),
        ("`ExprDensity` (term / pair expressions, hiprtc)", lambda: kmc.ExprDensity("-0.5 * x * x")),
        ("`CDensity` (function body, hiprtc)", lambda: kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return -0.5 * s;")),
        ("`HostLogPdf` (any Python callable)", lambda: kmc.HostLogPdf(lambda X: -0.5 * (X * X).sum(axis=1), vectorized=True)),
    ]


def theta(nw):
    return np.random.default_rng(3).standard_normal((nw, ND))


def _run(s, gens=70):
    s.run(gens)
    s.sync()
    assert s.generation == gens


def f_graph(kmc, pdf):
    with kmc.Sampler(pdf, 2048, ND, 200, 20, 1, 2.0, SEED, moments=True) as s:
        s.set_positions(theta(2048))
        _run(s, 140)
        mode, _ = s.launch_mode()
        if mode in (0, 1, 3):
            return "yes"
        return "no: " + ("a host round trip per half-step" if isinstance(pdf, kmc.HostLogPdf) else kmc.Sampler.LAUNCH_MODES[mode])


def f_resident(kmc, pdf):
    with kmc.Sampler(pdf, 256, ND, 200, 20, 1, 2.0, SEED, store_chain=True) as s:
        s.set_positions(theta(256))
        _run(s)
        return "yes" if "resident mode" in s.describe() else "no: runs in the multi-launch kernels"


def f_generation(kmc, pdf):
    with kmc.Sampler(pdf, 4096, ND, 200, 20, 1, 2.0, SEED, store_chain=True) as s:
        s.set_positions(theta(4096))
        _run(s)
        return "yes" if "one launch per generation" in s.describe() else "no: runs in the two-launch kernels"


def f_islands(kmc, pdf):
    with kmc.Sampler(pdf, 2048, ND, 200, 20, 1, 2.0, SEED, moments=True, island_gens=8, island_size=64) as s:
        s.set_positions(theta(2048))
        _run(s, 64)
        return "yes"


def f_f32(kmc, pdf):
    with kmc.Sampler(pdf, 2048, ND, 200, 20, 1, 2.0, SEED, store_chain=True, dtype="f32") as s:
        s.set_positions(theta(2048))
        _run(s)
        return "yes"


def f_stream(kmc, pdf):
    os.environ["KMC_DEBUG"] = "chain-block=8"
    try:
        with kmc.Sampler(pdf, 2048, ND, 100, 20, 1, 2.0, SEED, store_chain=True, store_logp=True, stream_chain=True, chain_by_walker=True) as s:
            s.set_positions(theta(2048))
            _run(s, 100)
            ch, lp = s.chain(by_walker=True)
            assert ch.shape == (2048, 80, ND)
            return "yes"
    finally:
        del os.environ["KMC_DEBUG"]


def _p2p(kmc, pdf, **kw):
    """Two shards of one ensemble in THIS process on one GPU (kmc_sampler_p2p_connect_local), driven like tests/test_gpu_p2p.py:
    the shards spin on each other's progress flags, so their streams must sit on different hardware queues (streams of
    different priority never share one) and only whole hipGraph chunks are enqueued."""
    import torch
    nw = 2048
    shards = [kmc.Sampler(pdf, nw, ND, 128, 10, 1, 2.0, SEED, moments=True, p2p=True, shard_rank=r, shard_count=2, **kw) for r in range(2)]
    streams = [torch.cuda.Stream(device=0, priority=-1), torch.cuda.Stream(device=0, priority=0)]
    try:
        for s, st in zip(shards, streams):
            s.set_stream(st.cuda_stream)
        kmc.Sampler.p2p_connect_local(shards)
        for s in shards:
            s.set_positions(theta(nw))
        for s in shards:
            s.run(128)
        for s in shards:
            s.sync()
        return shards[0].describe()
    finally:
        for s in shards:
            s.close()


def f_p2p_pull(kmc, pdf):
    _p2p(kmc, pdf)
    return "yes"


def f_allgather(kmc, pdf):
    with kmc.Sampler(pdf, 2048, ND, 200, 20, 1, 2.0, SEED, moments=True, shard_rank=0, shard_count=1) as s:
        s.rccl_init(kmc.Sampler.rccl_unique_id())
        captured = s.rccl_capture()
        s.set_positions(theta(2048))
        _run(s, 70)
        return "yes" + (" (all-gathers captured in the graph)" if captured else " (launch by launch)")


def f_dealt(kmc, pdf):
    with kmc.Sampler(pdf, 2048, ND, 200, 20, 1, 2.0, SEED, moments=True, deal_rank=0, deal_count=2) as s:
        s.set_positions(theta(2048))
        _run(s, 64)
        assert len(s.walker_ids()) == 2048
        return "yes"


def f_set_state(kmc, pdf):
    with kmc.Sampler(pdf, 2048, ND, 200, 20, 1, 2.0, SEED, moments=True) as s:
        s.set_positions(theta(2048))
        _run(s, 30)
        st = s.state()
    with kmc.Sampler(pdf, 2048, ND, 200, 20, 1, 2.0, SEED, moments=True) as s:
        s.restore(st)
        s.run(10)
        s.sync()
        assert s.generation == 40
        return "yes"


def f_init_ball(kmc, pdf):
    with kmc.Sampler(pdf, 2048, ND, 200, 20, 1, 2.0, SEED) as s:
        s.init_ball(np.zeros(ND), 0.1, seed=5)
        _run(s, 10)
        return "yes"


def f_blobs(kmc, pdf):
    from kissmcmc_jl_amd import api
    th = list(theta(64))
    if isinstance(pdf, kmc.HostLogPdf):
        fn = lambda x: (-0.5 * float(np.dot(x, x)), float(x[0]))
        r = api.emcee(fn, th, niter=64 * 20, hasblob=True, use_progress_meter=False, seed=1)
    elif isinstance(pdf, kmc.CDensity):      # a body that also fills blob[0..m): the blob is computed and carried on the device
        pdf = kmc.CDensity("double s = 0.0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; blob[0] = x[0]; return -0.5 * s;", nblob=1)
        r = api.emcee(pdf, th, niter=64 * 20, hasblob=True, use_progress_meter=False, seed=1)
        return "yes (`nblob=m`: m doubles per walker, on the device)" if np.array_equal(r[3][:, :, 0], r[0][:, :, 0]) else "no: blobs differ"
    else:
        r = api.emcee(pdf, th, niter=64 * 20, hasblob=True, use_progress_meter=False, seed=1)
    assert r[3] is not None and len(r[3]) == 64
    return "yes"


FEATURES = [
    ("hipGraph replay", f_graph),
    ("resident mode (≤ 1024 walkers; ≤ 2048 with ndim ≤ 8)", f_resident),
    ("one launch per generation (small states: ndim ≤ 8 up to 49 152 walkers and 196 608 doubles of state, longer rows up to 8 MiB of state with up to 49 152 walkers)", f_generation),
    ("islands (`KMC_ISLANDS`)", f_islands),
    ("float rows (`KMC_F32`)", f_f32),
    ("streamed chain (`KMC_STREAM_CHAIN`)", f_stream),
    ("P2P pull (`KMC_P2P`)", f_p2p_pull),
    ("RCCL all-gather shards", f_allgather),
    ("dealt sub-ensembles", f_dealt),
    ("`set_state` (resume)", f_set_state),
    ("`init_ball` (device `make_theta0s`)", f_init_ball),
    ("blobs (`hasblob=true`)", f_blobs),
]


def shorten(msg: str) -> str:
    msg = " ".join(str(msg).split())
    return msg if len(msg) <= 170 else msg[:167] + "..."


def cell(kmc, make_pdf, fn):
    """One cell; anything but a library refusal (KmcError) or a documented NotImplementedError propagates: a crash is a bug."""
    try:
        return fn(kmc, make_pdf())
    except kmc.KmcError as e:
        return f"no: {shorten(e)}"
    except NotImplementedError as e:
        return f"no: {shorten(e)}"


def matrix(kmc):
    rows = densities(kmc)
    return [(fname, [cell(kmc, mk, fn) for _, mk in rows]) for fname, fn in FEATURES], [n for n, _ in rows]


def render(kmc) -> str:
    body, heads = matrix(kmc)
    lines = ["| feature \\ log-density | " + " | ".join(heads) + " |", "|---|" + "---|" * len(heads)]
    for fname, cells in body:
        lines.append("| " + fname + " | " + " | ".join(c.replace("|", "\\|") for c in cells) + " |")
    return "\n".join(lines)


def readme_block() -> str:
    txt = open(os.path.join(ROOT, "README.md")).read()
    m = re.search(re.escape(BEGIN) + r"\n(.*?)\n" + re.escape(END), txt, re.S)
    return m.group(1) if m else None


def main():
    if "--write-from" in sys.argv:       # a table generated on the GPU box (gpurun_out/...), pasted into README.md here
        # (only the table's own lines: RCCL prints a version banner on stdout when a communicator is created)
        table = "\n".join(l for l in open(sys.argv[sys.argv.index("--write-from") + 1]).read().splitlines() if l.startswith("|"))
    else:
        import kissmcmc_jl_amd as kmc
        table = render(kmc)
        print(table)
    if "--write" in sys.argv or "--write-from" in sys.argv:
        p = os.path.join(ROOT, "README.md")
        txt = open(p).read()
        assert BEGIN in txt and END in txt, "README.md lacks the feature-matrix markers"
        txt = re.sub(re.escape(BEGIN) + r"\n.*?\n" + re.escape(END), lambda _: BEGIN + "\n" + table + "\n" + END, txt, flags=re.S)
        open(p, "w").write(txt)
        print("README.md updated", file=sys.stderr)


if __name__ == "__main__":
    main()
