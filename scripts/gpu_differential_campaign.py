#!/usr/bin/env python3
"""Randomised differential run of the HIP path against the CPU oracle on the GPU box (the
piles of oracle/differential_campaign.py: seed lengths 2.5-12 kb, depth 4-60x, error
1-25 %, heterozygosity, unrelated reads, low-complexity seeds, min_cov 0-8, min_idt
0.60-0.95).  Checker = oracle/libfalcon_oracle.so (test infrastructure); one batch per
(min_cov, min_idt) setting, consensus and eqv compared pile by pile.

    python scripts/gpu_differential_campaign.py 0 16      # seeds; 12 piles each
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(seed):
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    rng = random.Random(seed)
    out = []
    for t in range(12):
        S = rng.choice([2500, 3000, 4000, 7000, 12000])
        cov = rng.choice([4, 8, 15, 30, 60])
        e = rng.choice([0.01, 0.05, 0.10, 0.13, 0.18, 0.25])
        het = rng.choice([0, 0, 0.005, 0.03])
        mr = rng.choice([1500, 2500, 5000])
        s, rd = make_pile(seed * 1000 + t, S=S, coverage=cov, e=e, het=het, min_read=mr // 2,
                          mean_read=mr, sd_read=mr // 3)
        pile = [codes_to_str(x) for x in pile_to_seqs(s, rd, rng.choice([10, 60, 200, 500]))]
        if rng.random() < 0.3:
            for _ in range(rng.randint(1, 5)):
                pile.insert(rng.randint(1, len(pile)),
                            "".join(rng.choice("ACGT") for _ in range(rng.randint(50, 3000))))
        if rng.random() < 0.2:
            h = len(pile[0]) // 2
            pile[0] = pile[0][:h] + "AC" * 200 + pile[0][h:]
        out.append((pile, rng.choice([0, 2, 4, 8]), rng.choice([0.60, 0.70, 0.85, 0.95])))
    return out


def main():
    from falcon_amd.engine import Engine
    from oracle.pyoracle import Port
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    cases = [c for s in range(lo, hi) for c in make(s)]
    groups = {}
    for i, (pile, mc, idt) in enumerate(cases):
        groups.setdefault((mc, idt), []).append(i)
    eng, port = Engine(0), Port()
    bad, t0 = [], time.time()
    for (mc, idt), idx in sorted(groups.items()):
        got = eng.consensus([cases[i][0] for i in idx], mc, 8, idt, want_eqv=True)
        for i, g in zip(idx, got):
            if tuple(g) != tuple(port.generate_consensus(cases[i][0], mc, 8, idt)):
                bad.append((lo + i // 12, i % 12, mc, idt))
    eng.close()
    print("%d piles in %.1f s, mismatches: %s" % (len(cases), time.time() - t0, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
