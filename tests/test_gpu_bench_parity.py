"""bench.py's own parity leg at batch scale, in the driver-run suite: the D. melanogaster-like
and the Arabidopsis-like workloads (longer seeds and the
200-read cap binding; two haplotypes) as batches of 384 piles -- the GPU's consensus of 256
of them compared, string for string, with what the COMPILED REFERENCE (oracle/_ref, or the
restatement where the reference build did not travel) makes of the same piles on the host's
cores.  One step, no end-to-end leg: this is a correctness test, the numbers are not used."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["dmel", "arab"])
def test_bench_self_parity_at_batch_scale(workload):
    cores = os.cpu_count() or 1
    procs = max(4, min(128, cores // 2))
    timed = max(1, -(-256 // procs))           # >= 256 timed piles in all
    piles = procs * (timed + 1)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--piles", str(piles),
           "--steps", "1", "--warmup", "0", "--no-end-to-end", "--cpu-baseline-procs", str(procs),
           "--cpu-baseline-timed", str(timed)]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=1500,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.split("\n") if ln.startswith("{")][-1])
    assert line["config"]["piles_per_step_per_gpu"] == piles
    assert line["parity_checked_piles"] >= 256, line.get("cpu_baseline")
    assert line["parity_mismatches"] == 0, line.get("parity_mismatching_piles")
    # (against the reference's own C, built by oracle/Makefile and shipped as oracle/_ref: a run
    # that fell back to the repo's restatement would not be the check this test is named for)
    assert line["parity_against"] == "reference", line["parity_against"]
    assert line["align"]["handed_back"] <= line["config"]["sequences_per_step_per_gpu"] // 50
