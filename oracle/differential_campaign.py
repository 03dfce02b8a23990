#!/usr/bin/env python3
"""Long randomised differential run: the CPU restatement (oracle/falcon_oracle.c) against
the compiled reference (oracle/_ref).  TEST INFRASTRUCTURE ONLY, dev container only.

    python oracle/differential_campaign.py piles 0 64     # 12 piles per seed
    python oracle/differential_campaign.py functions 0 64 # 40 (q, t) pairs per seed

tests/test_oracle_vs_ref.py runs a short version of the same comparisons in the suite;
this is the long one (round 1: 864 piles and 2560 function-level cases, no mismatch).
Every seed runs in its own process: the reference is not re-entrant (falcon.c:338) and
abort()s on some failures, which must not take the campaign down silently."""
from __future__ import annotations

import multiprocessing as mp
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def piles(seed):
    from oracle.pyoracle import Port, Ref
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    rng = random.Random(seed)
    port, ref = Port(), Ref()
    bad = []
    for t in range(12):
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
        if tuple(port.generate_consensus(pile, mc, 8, idt)) != tuple(ref.generate_consensus(pile, mc, 8, idt)):
            bad.append(("pile", seed, t, S, cov, e, het, mc, idt))
    return bad


def functions(seed):
    import numpy as np
    from oracle.pyoracle import Port, Ref
    from falcon_amd.synth import codes_to_str, noisy
    rng, nrng = random.Random(seed), np.random.default_rng(seed)
    port, ref = Port(), Ref()
    bad = []
    for t in range(40):
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
        a, b = port.align(q, tt, band, 1), ref.align(q, tt, band, 1)
        a.pop("cells", None)
        b.pop("cells", None)
        if a != b:
            bad.append(("align", seed, t, n, e1, e2, band))
        hq, ht = port.find_hits(tt, q)
        if (hq, ht) != ref.find_hits(tt, q):
            bad.append(("hits", seed, t))
        for bs, th in ((48, 5), (80, 50), (16, 1)):
            if list(port.best_range(hq, ht, bs, th)) != list(ref.best_range(hq, ht, bs, th)):
                bad.append(("range", seed, t, bs, th))
        if list(port.best_range2(hq, ht)) != list(ref.best_range2(hq, ht)):
            bad.append(("range2", seed, t))
        for m in (4, 16):
            if port.find_hits(tt, q, 8, m) != ref.find_hits(tt, q, 8, m):
                bad.append(("masked hits", seed, t, m))
    return bad


def _one(job):
    kind, seed = job
    return {"piles": piles, "functions": functions}[kind](seed)


def main():
    kind, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    bad, dead = [], []
    ctx = mp.get_context("fork")
    with ctx.Pool(min(8, os.cpu_count() or 1), maxtasksperchild=1) as pool:
        pending = [(s, pool.apply_async(_one, ((kind, s),))) for s in range(lo, hi)]
        for s, r in pending:
            try:
                bad += r.get(timeout=1800)
            except Exception as exc:  # a worker that died (abort) or hung
                dead.append((s, repr(exc)))
    print("%s: seeds %d..%d, mismatches %d %s, seeds without a result %s"
          % (kind, lo, hi - 1, len(bad), bad[:10], dead))
    sys.exit(1 if bad or dead else 0)


if __name__ == "__main__":
    main()
