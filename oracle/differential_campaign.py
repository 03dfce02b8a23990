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
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def piles(seed):
    from oracle.campaign_cases import pile_cases
    from oracle.pyoracle import Port, Ref
    port, ref = Port(), Ref()
    bad = []
    for t, (pile, mc, idt) in enumerate(pile_cases(seed)):
        if tuple(port.generate_consensus(pile, mc, 8, idt)) != tuple(ref.generate_consensus(pile, mc, 8, idt)):
            bad.append(("pile", seed, t, len(pile[0]), len(pile), mc, idt))
    return bad


def functions(seed):
    from oracle.campaign_cases import function_cases
    from oracle.pyoracle import Port, Ref
    port, ref = Port(), Ref()
    bad = []
    for t, (q, tt, band) in enumerate(function_cases(seed)):
        a, b = port.align(q, tt, band, 1), ref.align(q, tt, band, 1)
        a.pop("cells", None)
        b.pop("cells", None)
        if a != b:
            bad.append(("align", seed, t, len(q), len(tt), band))
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
