"""Synthetic LA4Falcon-style piles (SURVEY.md section 8d / BASELINE.md section 3).

One pile = a seed read plus the raw reads overlapping it, exactly what
``LA4Falcon -H<cutoff> -fo`` streams into ``falcon_kit.mains.consensus``
(reference: falcon_kit/mains/consensus_task.py:90, consensus.py:161-209).

Model: random genome of 3*S bp, seed = noisy(genome[S:2S]);
reads of length max(3000, gauss(12000, 4000)) placed so that they overlap the
seed window by >= 1000 bp, clipped to the window (LA4Falcon only emits the
overlapping part), each passed through the noise model, until the summed read
bases reach coverage*S.  Noise per base, total rate e: deletion 0.25e,
substitution 0.15e, insertion-after 0.60e (uniform random base).

Everything is numpy-vectorised and seeded (PCG64) so a pile is a pure function
of (seed, S, coverage, e, het).
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def noisy(codes: np.ndarray, rng: np.random.Generator, e: float) -> np.ndarray:
    """Apply the 8d noise model to an array of base codes 0..3."""
    n = codes.shape[0]
    r = rng.random(n)
    p_del, p_sub = 0.25 * e, 0.15 * e
    is_del = r < p_del
    is_sub = (r >= p_del) & (r < p_del + p_sub)
    is_ins = (r >= p_del + p_sub) & (r < e)
    out_base = np.where(is_sub, (codes + rng.integers(1, 4, n, dtype=np.uint8)) & 3, codes)
    emit = (~is_del).astype(np.int64) + is_ins.astype(np.int64)
    total = int(emit.sum())
    src = np.repeat(np.arange(n), emit)
    out = out_base[src]
    # the inserted base is the second emission of a source base (or the only
    # one when that base was also ... it cannot be: del and ins are exclusive)
    first = np.ones(total, dtype=bool)
    first[1:] = src[1:] != src[:-1]
    n_ins = int((~first).sum())
    out[~first] = rng.integers(0, 4, n_ins, dtype=np.uint8)
    return out.astype(np.uint8)


def make_pile(seed: int, S: int = 20000, coverage: float = 40.0, e: float = 0.13,
              het: float = 0.0, min_read: int = 3000, mean_read: float = 12000.0,
              sd_read: float = 4000.0):
    """Return (seed_codes, [read_codes...]) as uint8 arrays of codes 0..3."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    genome = rng.integers(0, 4, 3 * S, dtype=np.uint8)
    haps = [genome]
    if het > 0.0:
        alt = genome.copy()
        m = rng.random(3 * S) < het
        alt[m] = (alt[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
        haps.append(alt)
    seed_read = noisy(genome[S:2 * S], rng, e)
    reads = []
    total = 0
    target = coverage * S
    rejected = 0
    while total < target:
        L = int(max(min_read, rng.normal(mean_read, sd_read)))
        # read [start, start+L) must overlap window [S, 2S) by >= 1000 bp
        start = int(rng.integers(S - L + 1000, 2 * S - 1000 + 1))
        lo, hi = max(start, S), min(start + L, 2 * S)
        if hi - lo < 1000:
            rejected += 1
            if rejected > 100000 and not reads:
                raise ValueError("no read of this length distribution overlaps the seed window by 1000 bp")
            continue
        hap = haps[int(rng.integers(0, len(haps)))] if len(haps) > 1 else genome
        rd = noisy(hap[lo:hi], rng, e)
        reads.append(rd)
        total += rd.shape[0]
    return seed_read, reads


def codes_to_str(codes: np.ndarray) -> str:
    return _ACGT[codes].tobytes().decode("ascii")


def pile_to_seqs(seed_read: np.ndarray, reads, max_n_read: int = 200):
    """Order a pile the way consensus.py hands it to C: seed, then seed copy and
    reads stably sorted by descending length, capped at max_n_read entries
    (reference: consensus.py:183-190 seed twice, :26-45 get_longest_reads)."""
    rest = [seed_read] + list(reads)
    rest = sorted(rest, key=lambda a: -a.shape[0])  # python sort is stable
    seqs = [seed_read] + rest
    return seqs[:max_n_read]


def pile_to_la4falcon(seed_id: str, seed_read: np.ndarray, reads, first_read_id: int = 1) -> str:
    """LA4Falcon text for one pile: '<id> <seq>' lines then '+ +'.
    (The seed line comes first; consensus.py adds the seed a second time itself.)"""
    lines = ["%s %s" % (seed_id, codes_to_str(seed_read))]
    for i, rd in enumerate(reads):
        lines.append("%08d %s" % (first_read_id + i, codes_to_str(rd)))
    lines.append("+ +")
    return "\n".join(lines) + "\n"
