#!/usr/bin/env python3
"""Digests of the COMPILED REFERENCE's answers (oracle/_ref) on the campaign cases of
oracle/campaign_cases.py -> tests/golden/f9_campaign.json.gz.  Dev container only (needs
oracle/_ref, i.e. /root/reference); TEST INFRASTRUCTURE ONLY.

    python oracle/gen_campaign_digests.py 72 64      # pile seeds 0..71, function seeds 0..63

The fixture holds no sequences: the cases are pure functions of their seed
(campaign_cases.py), the fixture the 20-hex-digit digests of (consensus, eqv) per pile and
of the six integers + two strings per alignment, plus a digest of each case's input so a
drifting generator is noticed before a false alarm is raised."""
from __future__ import annotations

import gzip
import hashlib
import json
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _piles(seed):
    from oracle.campaign_cases import consensus_digest, pile_cases
    from oracle.pyoracle import Ref
    ref = Ref()
    out = []
    for pile, mc, idt in pile_cases(seed):
        seq, eqv = ref.generate_consensus(pile, mc, 8, idt)
        out.append([hashlib.sha1("\n".join(pile).encode()).hexdigest()[:12],
                    consensus_digest(seq, eqv), len(seq)])
    return out


def _pairs(seed):
    from oracle.campaign_cases import align_digest, function_cases
    from oracle.pyoracle import Ref
    ref = Ref()
    out = []
    for q, t, band in function_cases(seed):
        a = ref.align(q, t, band, 1)
        out.append([hashlib.sha1((q + " " + t).encode()).hexdigest()[:12], align_digest(a),
                    a["aln_str_size"]])
    return out


def _one(job):
    return {"piles": _piles, "pairs": _pairs}[job[0]](job[1])


def main():
    n_pile, n_pair = int(sys.argv[1]), int(sys.argv[2])
    ctx = mp.get_context("fork")
    with ctx.Pool(os.cpu_count() or 1, maxtasksperchild=1) as pool:
        jobs = [("piles", s) for s in range(n_pile)] + [("pairs", s) for s in range(n_pair)]
        res = pool.map(_one, jobs, chunksize=1)
    doc = {"what": "compiled reference (oracle/_ref) on oracle/campaign_cases.py",
           "piles": res[:n_pile], "pairs": res[n_pile:]}
    path = os.path.join(ROOT, "tests", "golden", "f9_campaign.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(doc, separators=(",", ":")).encode())
    print("wrote %s: %d piles, %d pairs" % (path, sum(map(len, doc["piles"])),
                                            sum(map(len, doc["pairs"]))))


if __name__ == "__main__":
    main()
