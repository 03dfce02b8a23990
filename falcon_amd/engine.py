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


class FailedPile(str):
    """The (empty) consensus of a pile that failed alone (fa_batch_pile_error): its batch went
    on without it; ``reason`` says why.  Prints like a pile without usable reads: nothing."""

    def __new__(cls, reason):
        self = super().__new__(cls, "")
        self.reason = reason
        return self


class PileSet:
    """Admitted piles of one ``fa_reader_next`` call: pointer arrays owned by the reader
    (valid through its next call, not the one after), in exactly the shape
    ``fa_batch_create`` takes."""

    def __init__(self, n_pile, pile_n_seq, seqs, seq_len, seed_ids):
        self.n_pile = n_pile
        self.pile_n_seq, self.seqs, self.seq_len, self.raw_ids = pile_n_seq, seqs, seq_len, seed_ids
        self.seed_ids_raw = [seed_ids[i] for i in range(n_pile)]  # (bytes, copied out of the reader)
        self.seed_ids = [b.decode("ascii", "replace") for b in self.seed_ids_raw]
        self.first = [0] * (n_pile + 1)
        for p in range(n_pile):
            self.first[p + 1] = self.first[p] + pile_n_seq[p]

    def piles(self):
        """Materialise as python lists of str (tests, --trim)."""
        out = []
        for p in range(self.n_pile):
            out.append([C.string_at(self.seqs[g], self.seq_len[g]).decode("ascii")
                        for g in range(self.first[p], self.first[p + 1])])
        return out


class Reader:
    """Native LA4Falcon stream reader (falcon_amd/csrc/reader.cpp): grammar, pile
    admission and read selection of consensus.py:161-209 and :26-45."""

    def __init__(self, fd, min_n_read, min_len_aln, min_cov_aln, max_n_read, max_cov_aln):
        self.lib = load()
        self.h = self.lib.fa_reader_open(fd, min_n_read, min_len_aln, min_cov_aln, max_n_read,
                                         max_cov_aln)
        if not self.h:
            raise FalconAmdError("fa_reader_open failed")

    def keep(self, n_batches: int):
        """What next() hands out lives through the next ``n_batches`` - 1 calls (default 2)."""
        if self.lib.fa_reader_keep(self.h, int(n_batches)):
            raise FalconAmdError("fa_reader_keep(%d) refused" % n_batches)

    def next(self, max_piles=0, max_bases=0):
        """The next PileSet, or None at the end of the stream."""
        cnt = C.POINTER(C.c_int)()
        seqs = C.POINTER(C.c_char_p)()
        lens = C.POINTER(C.c_int)()
        ids = C.POINTER(C.c_char_p)()
        # (c_char_p arrays are only indexed for the NUL-terminated ids; the sequence
        # pointers are passed through as raw addresses)
        seqs_raw = C.cast(C.pointer(seqs), C.POINTER(C.POINTER(C.c_char_p)))
        n = self.lib.fa_reader_next(self.h, max_piles, max_bases, C.byref(cnt), seqs_raw,
                                    C.byref(lens), C.byref(ids))
        if n < 0:
            raise FalconAmdError(self.lib.fa_reader_error(self.h).decode("utf-8", "replace"))
        if n == 0:
            return None
        return PileSet(n, cnt, C.cast(seqs, C.POINTER(C.c_void_p)), lens, ids)

    def __iter__(self):
        while True:
            ps = self.next()
            if ps is None:
                return
            for sid, pile in zip(ps.seed_ids, ps.piles()):
                yield sid, pile

    def close(self):
        if self.h:
            self.lib.fa_reader_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """A set of piles resident in HBM (2-bit packed)."""

    @classmethod
    def from_pileset(cls, engine: "Engine", ps: PileSet, p0: int = 0, p1: int = None) -> "Batch":
        """Stage piles [p0, p1) of a PileSet: pointer arithmetic only, no per-sequence
        python objects."""
        p1 = ps.n_pile if p1 is None else p1
        self = cls.__new__(cls)
        self.engine, self.lib = engine, engine.lib
        self.n_pile = p1 - p0
        g0 = ps.first[p0]
        self.n_seq = ps.first[p1] - g0
        self.first = [ps.first[p] - g0 for p in range(p0, p1)]
        cnt = C.cast(C.addressof(ps.pile_n_seq.contents) + 4 * p0, C.POINTER(C.c_int))
        arr = C.cast(C.addressof(ps.seqs.contents) + C.sizeof(C.c_void_p) * g0, C.POINTER(C.c_char_p))
        lens = C.cast(C.addressof(ps.seq_len.contents) + 4 * g0, C.POINTER(C.c_int))
        self.h = self.lib.fa_batch_create(engine.h, self.n_pile, cnt, arr, lens)
        if not self.h:
            raise FalconAmdError(last_error())
        return self

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

    def submit(self, min_cov: int = 6, K: int = 8, min_idt: float = 0.70) -> "Batch":
        """First half of ``run``: queues the batch's seed index, chaining and alignment kernels
        and returns at once (it never waits for the device); the consensus stage follows on a
        stream of its own, driven by a thread of the engine's context.  Keep two or three
        batches of an engine submitted ahead of every ``wait`` to overlap them."""
        if self.lib.fa_batch_submit(self.h, min_cov, K, min_idt):
            raise FalconAmdError(last_error())
        return self

    def wait(self) -> "Batch":
        if self.lib.fa_batch_wait(self.h):
            raise FalconAmdError(last_error())
        return self

    def trim_windows(self, K: int = 8, mask_threshold: int = 16) -> "Batch":
        """--trim: find_best_aln_range2 of every read on its seed (results: ``range(g)``)."""
        if self.lib.fa_batch_trim_windows(self.h, K, mask_threshold):
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
        if n.value == 0:
            msg = C.create_string_buffer(256)
            if self.lib.fa_batch_pile_error(self.h, pile, msg, 256) > 0:
                s = FailedPile(msg.value.decode("utf-8", "replace"))
        return (s, list(eqv[:n.value])) if self._eqv else s

    def fasta(self, seed_ids, mode: int) -> bytes:
        """FASTA text of the fetched batch under the worker's output rules (consensus.py:275-299;
        mode 0 default, 1 --output-multi, 2 --output-full), failed piles left out.
        ``seed_ids``: bytes (or str), one per pile."""
        ids = (C.c_char_p * self.n_pile)(*[s if isinstance(s, bytes) else s.encode("utf-8") for s in seed_ids])
        text = C.c_void_p()
        n = C.c_longlong()
        if self.lib.fa_batch_fasta(self.h, ids, mode, C.byref(text), C.byref(n)):
            raise FalconAmdError(last_error())
        return C.string_at(text, n.value)

    def failures(self):
        """[(pile, reason)] of the piles without a consensus (fa_batch_pile_error)."""
        out = []
        if self.stats().n_piles_failed:
            msg = C.create_string_buffer(256)
            for p in range(self.n_pile):
                if self.lib.fa_batch_pile_error(self.h, p, msg, 256) > 0:
                    out.append((p, msg.value.decode("utf-8", "replace")))
        return out

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

    def debug_hits(self, g: int):
        """(query_pos[], target_pos[]) of sequence g's k-mer hits on its seed, in the order the
        chain stage enumerates them (tests)."""
        n = self.lib.fa_batch_debug_hits(self.h, g, None, None, 0)
        if n < 0:
            raise FalconAmdError(last_error())
        q, t = (C.c_int * max(n, 1))(), (C.c_int * max(n, 1))()
        if self.lib.fa_batch_debug_hits(self.h, g, q, t, n) != n:
            raise FalconAmdError(last_error())
        return list(q[:n]), list(t[:n])

    def debug_tags(self, g: int):
        """The tag words of sequence g's accepted alignment decoded into the reference's tag
        list [(t_pos relative to the alignment's first seed position, delta, base or '-' or
        None)] -- base None: the read has the seed's base there (tests); [] when the alignment
        was not accepted."""
        lead, n_ins = C.c_uint(), C.c_int()
        n = self.lib.fa_batch_debug_tags(self.h, g, None, 0, C.byref(lead), None, 0, C.byref(n_ins))
        if n < 0:
            raise FalconAmdError(last_error())
        if n == 0:
            return []
        words = (C.c_uint * n)()
        ins = (C.c_ubyte * max(n_ins.value, 1))()
        if self.lib.fa_batch_debug_tags(self.h, g, words, n, C.byref(lead), ins, n_ins.value, C.byref(n_ins)) != n:
            raise FalconAmdError(last_error())

        def run(word, t):
            k = (word >> 23) & 0xff
            pay = word & 0x7fffff
            for d in range(k):
                code = (pay >> (2 * d)) & 3 if k <= 11 else ins[pay + d]
                yield (t, d + 1, "ACGT"[code])

        out = list(run(lead.value, -1))
        for t in range(n):
            out.append((t, 0, "-" if words[t] >> 31 else None))
            out.extend(run(words[t], t))
        return out

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

    def warm(self, batch_bases: int):
        """Pay now what the first batch would pay for (pinned staging, code objects, first upload)."""
        if self.lib.fa_warm(self.h, int(batch_bases)):
            raise FalconAmdError(last_error())

    def batch(self, piles) -> Batch:
        return Batch(self, piles)

    def consensus(self, piles, min_cov=6, K=8, min_idt=0.70, want_eqv=False):
        """Consensus of every pile (list of lists of sequences, or a (PileSet, p0, p1)
        slice straight from the native reader), in order."""
        b = Batch.from_pileset(self, *piles) if isinstance(piles, tuple) else self.batch(piles)
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
