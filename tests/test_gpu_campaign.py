"""Slices of the randomised differential campaign (oracle/campaign_cases.py) in the suite.

The expected answers are the digests of tests/golden/f9_campaign.json.gz, produced in the
dev container by the COMPILED REFERENCE (oracle/gen_campaign_digests.py).  CPU: the
restatement against them (pins fixture and generator); GPU: the HIP path through the C
ABI against them -- unrelated reads, low-complexity seeds, 1-25 % error, min_cov 0-8,
min_idt 0.60-0.95 (falcon.c:597-647) and `align` over bands 10-1500 with tails and
truncations (DW_banded.c:183-243).  scripts/gpu_differential_campaign.py runs all of it."""
import hashlib
import os
import sys

import pytest

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_campaign_fixture_vs_restatement(port):
    from oracle.campaign_cases import align_digest, consensus_digest, function_cases, pile_cases
    d = load_golden("f9_campaign")
    assert len(d["piles"]) >= 72 and len(d["pairs"]) >= 64
    for s in (3, 40):
        for t, (pile, mc, idt) in enumerate(pile_cases(s)):
            assert hashlib.sha1("\n".join(pile).encode()).hexdigest()[:12] == d["piles"][s][t][0]
            assert consensus_digest(*port.generate_consensus(pile, mc, 8, idt)) == d["piles"][s][t][1]
    for s in (5, 33):
        for t, (q, tt, band) in enumerate(function_cases(s)):
            a = port.align(q, tt, band, 1)
            a.pop("cells", None)
            assert align_digest(a) == d["pairs"][s][t][1], (s, t)


@pytest.mark.gpu
def test_campaign_piles_on_the_gpu():
    import gpu_differential_campaign as camp
    from falcon_amd.engine import Engine
    eng = Engine(0)
    try:
        # the WHOLE frozen campaign: 300 pile seeds x 12 shapes (11 s on the GPU box)
        n, bad, stale = camp.run_piles(eng, 0, 300, load_golden("f9_campaign"))
    finally:
        eng.close()
    assert n == 3600 and not stale and not bad, (bad, stale)


@pytest.mark.gpu
def test_campaign_pairs_on_the_gpu():
    import gpu_differential_campaign as camp
    from falcon_amd.engine import Engine
    eng = Engine(0)
    try:
        # ... and all 256 pair seeds x 40 shapes over bands 10..1500 (7 s)
        n, bad, stale = camp.run_pairs(eng, 0, 256, load_golden("f9_campaign"))
    finally:
        eng.close()
    assert n == 10240 and not stale and not bad, (bad, stale)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2"])
def test_the_hand_scheduled_rows_against_their_statement_on_the_device(mode):
    """k_align2's row loop is one hand-written gfx950 instruction stream (k_align2_rows.h) that since round 6 also
    lays the two bands out again inside the asm statement; what it computes is stated in portable code beside it
    (a2_rows_c + a2_replace, k_align2_core.h -- what the lane emulator runs).  k_align2_shadow runs BOTH from the
    same state, stretch by stretch, and logs every difference in what they hand back: iteration, masks, best_m,
    every lane's x, the tape's K fields, cell counts, the layout (split, zones, lane-0 diagonals), the number of
    re-layouts.  FALCON_AMD_A2_SHADOW=1 goes on with the statement's state, =2 with the stream's (as the product
    does).  Campaign pairs at band 150, bench-like piles and the frozen campaign's pile shapes: every launch must
    report 0 differing stretches, and the answers are checked against the campaign digests."""
    import subprocess
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "scripts")); sys.path.insert(0, os.path.join(%r, "tests"))
import gpu_differential_campaign as camp
from conftest import load_golden
from falcon_amd.engine import Engine
from benchlib.workloads import WORKLOADS, gen_piles
eng = Engine(0)
gold = load_golden("f9_campaign")
n, bad, stale = camp.run_pairs(eng, 0, 48, gold)
assert n == 48 * 40 and not bad and not stale, (n, bad, stale)
n, bad, stale = camp.run_piles(eng, 0, 24, gold)
assert not bad and not stale, (bad, stale)
piles = gen_piles(range(700, 716), 1, WORKLOADS["ecoli"])
out = eng.consensus(piles, 4, 8, 0.70)
assert all(len(s) > 15000 for s in out)
eng.close()
print("shadow run done")
''' % (ROOT, ROOT, ROOT)
    env = dict(os.environ, FALCON_AMD_A2_SHADOW=mode)
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0 and b"shadow run done" in p.stdout, err[-3000:]
    lines = [ln for ln in err.splitlines() if ln.startswith("a2_shadow:")]
    assert len(lines) >= 3, err[-2000:]                       # (the shadow kernel is what ran)
    assert all(ln.strip() == "a2_shadow: 0 stretches differ" for ln in lines), "\n".join(err.splitlines()[:60])
