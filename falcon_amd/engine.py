"""Python face of the batch ABI: one ``Engine`` per process and GPU.

Replaces the ``multiprocessing.Pool.imap`` over piles of the reference driver
(falcon_kit/mains/consensus.py:264-274, falcon_kit/multiproc.py:28-36): piles
are submitted in batches, every stage runs as HIP kernels over the whole batch,
and results come back in submission order.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

from .lib import Alignment, FalconAmdError, FaStats, last_error, load


def _b(s) -> bytes:
    return s if isinstance(s, bytes) else s.encode("ascii")


class Batch:
    """A set of piles resident in HBM (2-bit packed)."""

    def __init__(self, engine: "Engine", piles: Sequence[Sequence]):
        self.engine = engine
        self.lib = engine.lib
        flat: List[bytes] = []
        counts = []
        for pile in piles:
            counts.append(len(pile))
            flat.extend(_b(s) for s in pile)
        self.n_pile = len(counts)
        self.n_seq = len(flat)
        self.first = []
        g = 0
        for c in counts:
            self.first.append(g)
            g += c
        arr = (C.c_char_p * max(1, len(flat)))(*flat)
        lens = (C.c_int * max(1, len(flat)))(*[len(s) for s in flat])
        cnt = (C.c_int * max(1, len(counts)))(*counts)
        self.h = self.lib.fa_batch_create(engine.h, self.n_pile, cnt, arr, lens)
        if not self.h:
            raise FalconAmdError(last_error())

    def run(self, min_cov: int = 6, K: int = 8, min_idt: float = 0.70) -> "Batch":
        if self.lib.fa_batch_run(self.h, min_cov, K, min_idt):
            raise FalconAmdError(last_error())
        return self

    def fetch(self, want_eqv: bool = False) -> "Batch":
        if self.lib.fa_batch_fetch(self.h, 1 if want_eqv else 0):
            raise FalconAmdError(last_error())
        self._eqv = want_eqv
        return self

    def result(self, pile: int):
        seq = C.c_char_p()
        n = C.c_int()
        eqv = C.POINTER(C.c_int)()
        if self.lib.fa_batch_result(self.h, pile, C.byref(seq), C.byref(n),
                                    C.byref(eqv) if self._eqv else None):
            raise FalconAmdError(last_error())
        s = C.string_at(seq, n.value).decode("ascii")
        return (s, list(eqv[:n.value])) if self._eqv else s

    def stats(self) -> FaStats:
        st = FaStats()
        self.lib.fa_batch_stats(self.h, C.byref(st))
        return st

    def range(self, g: int):
        v = [C.c_int() for _ in range(4)]
        score = C.c_longlong()
        ok, nh = C.c_int(), C.c_int()
        if self.lib.fa_batch_range(self.h, g, *[C.byref(x) for x in v], C.byref(score),
                                   C.byref(ok), C.byref(nh)):
            raise FalconAmdError("fa_batch_range failed")
        return dict(s1=v[0].value, e1=v[1].value, s2=v[2].value, e2=v[3].value,
                    score=score.value, ok=ok.value, n_hit=nh.value)

    def alignment(self, g: int):
        v = [C.c_int() for _ in range(5)]
        cells = C.c_longlong()
        if self.lib.fa_batch_alignment(self.h, g, *[C.byref(x) for x in v], C.byref(cells)):
            raise FalconAmdError("fa_batch_alignment failed")
        return dict(dist=v[0].value, q_e=v[1].value, t_e=v[2].value, size=v[3].value,
                    accept=v[4].value, cells=cells.value)

    def free(self):
        if self.h:
            self.lib.fa_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, device: int = 0):
        self.lib = load()
        self.h = self.lib.fa_create(device)
        if not self.h:
            raise FalconAmdError(last_error())
        self.device = device

    def batch(self, piles) -> Batch:
        return Batch(self, piles)

    def consensus(self, piles, min_cov=6, K=8, min_idt=0.70, want_eqv=False):
        """Consensus of every pile (list of lists of sequences), in order."""
        b = self.batch(piles)
        try:
            b.run(min_cov, K, min_idt).fetch(want_eqv)
            return [b.result(p) for p in range(b.n_pile)]
        finally:
            b.free()

    def align_pairs(self, pairs, band=150, want_str=True):
        """Banded O(ND) alignment of (query, target) pairs in one launch."""
        n = len(pairs)
        q = [_b(a) for a, _ in pairs]
        t = [_b(b) for _, b in pairs]
        qa = (C.c_char_p * n)(*q)
        ta = (C.c_char_p * n)(*t)
        ql = (C.c_int * n)(*[len(x) for x in q])
        tl = (C.c_int * n)(*[len(x) for x in t])
        out = (C.POINTER(Alignment) * n)()
        if self.lib.fa_align_pairs(self.h, n, qa, ql, ta, tl, band, 1 if want_str else 0, out):
            raise FalconAmdError(last_error())
        res = []
        for i in range(n):
            r = out[i][0]
            res.append(dict(aln_str_size=r.aln_str_size, dist=r.dist, aln_q_s=r.aln_q_s,
                            aln_q_e=r.aln_q_e, aln_t_s=r.aln_t_s, aln_t_e=r.aln_t_e,
                            q_aln_str=C.string_at(r.q_aln_str).decode(),
                            t_aln_str=C.string_at(r.t_aln_str).decode()))
            self.lib.free_alignment(out[i])
        return res

    def close(self):
        if self.h:
            self.lib.fa_destroy(self.h)
            self.h = None
