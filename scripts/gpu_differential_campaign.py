#!/usr/bin/env python3
"""Randomised differential run of the HIP path on the GPU box against the COMPILED
REFERENCE's answers on the campaign cases (oracle/campaign_cases.py: seed lengths
2.5-12 kb, depth 4-60x, error 1-25 %, heterozygosity, unrelated reads, low-complexity
seeds, min_cov 0-8, min_idt 0.60-0.95; (q, t) pairs over bands 10-1500).  The answers are
the digests of tests/golden/f9_campaign.json.gz (oracle/gen_campaign_digests.py, made in
the dev container from oracle/_ref), so no CPU oracle time is spent on the GPU box.

    python scripts/gpu_differential_campaign.py piles 0 72     # 12 piles per seed
    python scripts/gpu_differential_campaign.py pairs 0 64     # 40 alignments per seed

Prints one summary line and the list of mismatching (seed, index) cases; exit 1 on any."""
import gzip
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_digests():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "f9_campaign.json.gz"), "rt") as f:
        return json.load(f)


def run_piles(eng, lo, hi, digests):
    """-> (n_cases, mismatches, stale inputs)"""
    from oracle.campaign_cases import consensus_digest, pile_cases
    cases, want = [], []
    for s in range(lo, hi):
        for t, c in enumerate(pile_cases(s)):
            cases.append((s, t) + c)
            want.append(digests["piles"][s][t])
    groups = {}
    for i, (s, t, pile, mc, idt) in enumerate(cases):
        groups.setdefault((mc, idt), []).append(i)
    bad, stale = [], []
    for (mc, idt), idx in sorted(groups.items()):
        got = eng.consensus([cases[i][2] for i in idx], mc, 8, idt, want_eqv=True)
        for i, (seq, eqv) in zip(idx, got):
            s, t, pile = cases[i][:3]
            if hashlib.sha1("\n".join(pile).encode()).hexdigest()[:12] != want[i][0]:
                stale.append((s, t))
            elif consensus_digest(seq, eqv) != want[i][1]:
                bad.append((s, t, mc, idt, len(seq), want[i][2]))
    return len(cases), bad, stale


def run_pairs(eng, lo, hi, digests):
    from oracle.campaign_cases import align_digest, function_cases
    cases, want = [], []
    for s in range(lo, hi):
        for t, c in enumerate(function_cases(s)):
            cases.append((s, t) + c)
            want.append(digests["pairs"][s][t])
    groups = {}
    for i, c in enumerate(cases):
        groups.setdefault(c[4], []).append(i)
    bad, stale = [], []
    for band, idx in sorted(groups.items()):
        got = eng.align_pairs([(cases[i][2], cases[i][3]) for i in idx], band, True)
        for i, a in zip(idx, got):
            s, t, q, tt = cases[i][:4]
            if hashlib.sha1((q + " " + tt).encode()).hexdigest()[:12] != want[i][0]:
                stale.append((s, t))
            elif align_digest(a) != want[i][1]:
                bad.append((s, t, band, a["aln_str_size"], want[i][2]))
    return len(cases), bad, stale


def main():
    from falcon_amd.engine import Engine
    kind, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    digests = load_digests()
    eng = Engine(0)
    t0 = time.time()
    n, bad, stale = {"piles": run_piles, "pairs": run_pairs}[kind](eng, lo, hi, digests)
    eng.close()
    print("%s seeds %d..%d: %d cases in %.1f s, mismatches vs the compiled reference: %d %s, "
          "cases whose input no longer matches the fixture: %d %s"
          % (kind, lo, hi - 1, n, time.time() - t0, len(bad), bad, len(stale), stale))
    sys.exit(1 if bad or stale else 0)


if __name__ == "__main__":
    main()
