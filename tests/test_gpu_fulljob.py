"""GPU: the four jobs of the bench line END TO END against the oracle's uninterrupted run, on the default planner -- C2 10^4 generations,
C3 10^4, C5 2 000, C1 1 000 (bench.py's exact inputs and seed): final positions and acceptance counters bit for bit, log-pdfs within
1e-12, moments within 1e-11 (`scripts/fulljob_parity.py: compare`, which also writes profiles/r06_fulljob_parity.txt when run by hand).
The reference's own long-run anchor is test/runtests.jl:68-72 (truths "from running emcee with niter=10^9"); the loop is
src/samplers.jl:245-293.  Oracle time on the GPU box's 16 threads: ~9 s (C2), ~3 s (C3), ~2 s (C5)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.fixture(scope="module")
def fulljobs(kmc, oracle):
    import fulljob_parity
    return fulljob_parity, fulljob_parity.jobs()


@pytest.mark.parametrize("name,kernel", [("C2", "half_step_vec"), ("C3", "generation_group"), ("C5", "half_step_vec"), ("C1", "resident")])
def test_whole_bench_job_ends_where_the_oracles_run_ends(fulljobs, name, kernel):
    mod, jobs = fulljobs
    ok, line, how = mod.compare(name, jobs[name])
    assert kernel in how, how              # (the default planner's kernel for this job: what the bench line times)
    assert ok, line
