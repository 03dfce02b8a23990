"""The randomised cases of the differential campaign.  TEST INFRASTRUCTURE ONLY.

One definition shared by the three users, so that "case (seed, t)" means the same input
everywhere:

* oracle/differential_campaign.py   -- restatement vs compiled reference, CPU
* oracle/gen_campaign_digests.py    -- digests of the compiled reference's answers
                                       -> tests/golden/f9_campaign.json.gz
* scripts/gpu_differential_campaign.py, tests/test_gpu_campaign.py
                                    -- HIP path vs those digests, GPU box

Pile cases cover what the parity fixtures do not: seed lengths 2.5-12 kb, depth 4-60x,
error 1-25 %, heterozygosity, unrelated reads, low-complexity seeds, min_cov 0-8,
min_idt 0.60-0.95 (domain of falcon.c:597-647).  Function cases are (q, t, band) pairs
for `align` over bands 10-1500 (DW_banded.c:183-243) with tails, prefixes and
truncations."""
from __future__ import annotations

import hashlib
import random

import numpy as np

PILES_PER_SEED = 12
PAIRS_PER_SEED = 40


def pile_cases(seed):
    """-> [(seqs, min_cov, min_idt)] * 12, a pure function of `seed`."""
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    rng = random.Random(seed)
    out = []
    for t in range(PILES_PER_SEED):
        S = rng.choice([2500, 3000, 4000, 7000, 12000])
        cov = rng.choice([4, 8, 15, 30, 60])
        e = rng.choice([0.01, 0.05, 0.10, 0.13, 0.18, 0.25])
        het = rng.choice([0, 0, 0.005, 0.03])
        mr = rng.choice([1500, 2500, 5000])
        s, rd = make_pile(seed * 1000 + t, S=S, coverage=cov, e=e, het=het, min_read=mr // 2,
                          mean_read=mr, sd_read=mr // 3)
        pile = [codes_to_str(x) for x in pile_to_seqs(s, rd, rng.choice([10, 60, 200, 500]))]
        if rng.random() < 0.3:   # unrelated reads
            for _ in range(rng.randint(1, 5)):
                pile.insert(rng.randint(1, len(pile)),
                            "".join(rng.choice("ACGT") for _ in range(rng.randint(50, 3000))))
        if rng.random() < 0.2:   # a low-complexity stretch in the seed
            h = len(pile[0]) // 2
            pile[0] = pile[0][:h] + "AC" * 200 + pile[0][h:]
        mc, idt = rng.choice([0, 2, 4, 8]), rng.choice([0.60, 0.70, 0.85, 0.95])
        out.append((pile, mc, idt))
    return out


def function_cases(seed):
    """-> [(q, t, band)] * 40, a pure function of `seed`."""
    from falcon_amd.synth import codes_to_str, noisy
    rng, nrng = random.Random(seed), np.random.default_rng(seed)
    out = []
    for t in range(PAIRS_PER_SEED):
        n = rng.choice([30, 200, 900, 2500, 6000])
        g = nrng.integers(0, 4, n, dtype=np.uint8)
        e1, e2 = rng.choice([0, 0.02, 0.08, 0.13, 0.2]), rng.choice([0, 0.05, 0.13, 0.25])
        q = codes_to_str(noisy(g, nrng, e1)) if e1 else codes_to_str(g)
        tt = codes_to_str(noisy(g, nrng, e2)) if e2 else codes_to_str(g)
        if rng.random() < 0.15:
            tt = tt[rng.randint(0, len(tt) // 3):]
        if rng.random() < 0.15:
            q += "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 400)))
        if rng.random() < 0.1:
            q = "A" * rng.randint(5, 300) + q
        band = rng.choice([10, 50, 150, 500, 1500])
        out.append((q, tt, band))
    return out


def consensus_digest(seq, eqv):
    h = hashlib.sha1(seq.encode("ascii"))
    h.update(np.asarray(eqv, dtype="<i4").tobytes())
    return h.hexdigest()[:20]


def align_digest(a):
    """a: dict as returned by pyoracle's align()."""
    s = "%d %d %d %d %d %d %s %s" % (a["aln_str_size"], a["dist"], a["aln_q_s"], a["aln_q_e"],
                                     a["aln_t_s"], a["aln_t_e"], a["q_aln_str"], a["t_aln_str"])
    return hashlib.sha1(s.encode("ascii")).hexdigest()[:20]
