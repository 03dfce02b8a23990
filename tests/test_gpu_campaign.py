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

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))


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
