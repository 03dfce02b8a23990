"""ctypes front-ends for the two CPU checkers.  TEST INFRASTRUCTURE ONLY.

* ``Port``  -- oracle/libfalcon_oracle.so, our C restatement (falcon_oracle.c).
* ``Ref``   -- oracle/_ref/falcon_ref.so, the real reference C path compiled by
  oracle/Makefile from /root/reference/src/c (ABI: src/c/common.h:57-177,
  falcon.c:562-566).  Present only if it was built in the dev container; it
  travels to the GPU box as a prebuilt artefact.

Both expose the same Python-level methods so tests can swap them:
``find_hits, best_range, best_range2, align, generate_consensus``.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "libfalcon_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "falcon_ref.so")


def build(quiet: bool = True) -> None:
    """(Re)build the oracle libraries with oracle/Makefile."""
    subprocess.run(["make", "-C", HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _b(s) -> bytes:
    return s if isinstance(s, bytes) else s.encode("ascii")


# ----------------------------------------------------------------------------
# Port (our restatement)
# ----------------------------------------------------------------------------
class _FoHits(C.Structure):
    _fields_ = [("count", C.c_int), ("query_pos", C.POINTER(C.c_int)),
                ("target_pos", C.POINTER(C.c_int))]


class _FoRange(C.Structure):
    _fields_ = [("s1", C.c_int), ("e1", C.c_int), ("s2", C.c_int), ("e2", C.c_int),
                ("score", C.c_long)]


class _FoAlignment(C.Structure):
    _fields_ = [("aln_str_size", C.c_int), ("dist", C.c_int), ("aln_q_s", C.c_int),
                ("aln_q_e", C.c_int), ("aln_t_s", C.c_int), ("aln_t_e", C.c_int),
                ("q_aln_str", C.c_void_p), ("t_aln_str", C.c_void_p), ("cells", C.c_long)]


class _FoConsensus(C.Structure):
    _fields_ = [("sequence", C.c_void_p), ("eqv", C.POINTER(C.c_int)),
                ("stat_L", C.c_long), ("stat_C", C.c_long), ("stat_D", C.c_long),
                ("stat_A", C.c_long), ("stat_T", C.c_long), ("stat_O", C.c_long),
                ("n_aligned", C.c_int)]


class Port:
    kind = "port"

    def __init__(self, path: str = PORT_SO):
        if not os.path.exists(path):
            build()
        self.lib = lib = C.CDLL(path)
        lib.fo_find_hits_masked.restype = C.POINTER(_FoHits)
        lib.fo_find_hits_masked.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                            C.c_int]
        lib.fo_free_hits.argtypes = [C.POINTER(_FoHits)]
        lib.fo_best_range.argtypes = [C.POINTER(_FoHits), C.c_int, C.c_int, C.POINTER(_FoRange)]
        lib.fo_best_range2.argtypes = [C.POINTER(_FoHits), C.POINTER(_FoRange)]
        lib.fo_align.restype = C.POINTER(_FoAlignment)
        lib.fo_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]
        lib.fo_free_alignment.argtypes = [C.POINTER(_FoAlignment)]
        lib.fo_generate_consensus.restype = C.POINTER(_FoConsensus)
        lib.fo_generate_consensus.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int,
                                              C.c_double]
        lib.fo_free_consensus.argtypes = [C.POINTER(_FoConsensus)]

    @staticmethod
    def _mk_hits(q, t):
        n = len(q)
        qa = (C.c_int * max(n, 1))(*q)
        ta = (C.c_int * max(n, 1))(*t)
        h = _FoHits(n, C.cast(qa, C.POINTER(C.c_int)), C.cast(ta, C.POINTER(C.c_int)))
        return h, (qa, ta)

    def find_hits(self, seed, query, K=8, mask=-1):
        seed, query = _b(seed), _b(query)
        h = self.lib.fo_find_hits_masked(seed, len(seed), query, len(query), K, mask)
        n = h[0].count
        out = (list(h[0].query_pos[:n]), list(h[0].target_pos[:n]))
        self.lib.fo_free_hits(h)
        return out

    def best_range(self, q, t, bin_size=48, count_th=5):
        h, _keep = self._mk_hits(q, t)
        r = _FoRange()
        self.lib.fo_best_range(C.byref(h), bin_size, count_th, C.byref(r))
        return (r.s1, r.e1, r.s2, r.e2, r.score)

    def best_range2(self, q, t):
        h, _keep = self._mk_hits(q, t)
        r = _FoRange()
        self.lib.fo_best_range2(C.byref(h), C.byref(r))
        return (r.s1, r.e1, r.s2, r.e2, r.score)

    def align(self, q, t, band=150, want_str=1):
        q, t = _b(q), _b(t)
        a = self.lib.fo_align(q, len(q), t, len(t), band, want_str)
        r = a[0]
        out = dict(aln_str_size=r.aln_str_size, dist=r.dist, aln_q_s=r.aln_q_s,
                   aln_q_e=r.aln_q_e, aln_t_s=r.aln_t_s, aln_t_e=r.aln_t_e,
                   q_aln_str=C.string_at(r.q_aln_str).decode(),
                   t_aln_str=C.string_at(r.t_aln_str).decode(), cells=r.cells)
        self.lib.fo_free_alignment(a)
        return out

    def generate_consensus(self, seqs, min_cov=4, K=8, min_idt=0.70, want_stats=False):
        seqs = [_b(s) for s in seqs]
        arr = (C.c_char_p * len(seqs))(*seqs)
        c = self.lib.fo_generate_consensus(arr, len(seqs), min_cov, K, min_idt)
        r = c[0]
        seq = C.string_at(r.sequence).decode()
        eqv = list(r.eqv[:len(seq)])
        stats = dict(L=r.stat_L, C=r.stat_C, D=r.stat_D, A=r.stat_A, T=r.stat_T, O=r.stat_O,
                     n_aligned=r.n_aligned)
        self.lib.fo_free_consensus(c)
        return (seq, eqv, stats) if want_stats else (seq, eqv)


# ----------------------------------------------------------------------------
# Ref (the reference's own C, ABI per src/c/common.h)
# ----------------------------------------------------------------------------
class _KmerLookup(C.Structure):
    _fields_ = [("start", C.c_int), ("last", C.c_int), ("count", C.c_int)]


class _KmerMatch(C.Structure):
    _fields_ = [("count", C.c_int), ("query_pos", C.POINTER(C.c_int)),
                ("target_pos", C.POINTER(C.c_int))]


class _AlnRange(C.Structure):
    _fields_ = [("s1", C.c_int), ("e1", C.c_int), ("s2", C.c_int), ("e2", C.c_int),
                ("score", C.c_long)]


class _Alignment(C.Structure):
    _fields_ = [("aln_str_size", C.c_int), ("dist", C.c_int), ("aln_q_s", C.c_int),
                ("aln_q_e", C.c_int), ("aln_t_s", C.c_int), ("aln_t_e", C.c_int),
                ("q_aln_str", C.c_void_p), ("t_aln_str", C.c_void_p)]


class _ConsensusData(C.Structure):
    _fields_ = [("sequence", C.c_void_p), ("eqv", C.POINTER(C.c_int))]


def bind_legacy_abi(lib):
    """argtypes/restype for the 19 symbols falcon_kit/falcon_kit.py:54-122 binds
    (with the true 5-argument generate_consensus of falcon.c:562-566)."""
    lib.allocate_kmer_lookup.restype = C.POINTER(_KmerLookup)
    lib.allocate_kmer_lookup.argtypes = [C.c_int]
    lib.init_kmer_lookup.argtypes = [C.POINTER(_KmerLookup), C.c_int]
    lib.free_kmer_lookup.argtypes = [C.POINTER(_KmerLookup)]
    lib.allocate_seq.restype = C.POINTER(C.c_uint8)
    lib.allocate_seq.argtypes = [C.c_int]
    lib.init_seq_array.argtypes = [C.POINTER(C.c_uint8), C.c_int]
    lib.free_seq_array.argtypes = [C.POINTER(C.c_uint8)]
    lib.allocate_seq_addr.restype = C.POINTER(C.c_int)
    lib.allocate_seq_addr.argtypes = [C.c_int]
    lib.free_seq_addr_array.argtypes = [C.POINTER(C.c_int)]
    lib.add_sequence.argtypes = [C.c_int, C.c_uint, C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                 C.POINTER(C.c_uint8), C.POINTER(_KmerLookup)]
    lib.mask_k_mer.argtypes = [C.c_int, C.POINTER(_KmerLookup), C.c_int]
    lib.find_kmer_pos_for_seq.restype = C.POINTER(_KmerMatch)
    lib.find_kmer_pos_for_seq.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.POINTER(C.c_int),
                                          C.POINTER(_KmerLookup)]
    lib.free_kmer_match.argtypes = [C.POINTER(_KmerMatch)]
    lib.find_best_aln_range.restype = C.POINTER(_AlnRange)
    lib.find_best_aln_range.argtypes = [C.POINTER(_KmerMatch), C.c_int, C.c_int, C.c_int]
    lib.find_best_aln_range2.restype = C.POINTER(_AlnRange)
    lib.find_best_aln_range2.argtypes = [C.POINTER(_KmerMatch), C.c_int, C.c_int, C.c_int]
    lib.free_aln_range.argtypes = [C.POINTER(_AlnRange)]
    lib.align.restype = C.POINTER(_Alignment)
    lib.align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]
    lib.free_alignment.argtypes = [C.POINTER(_Alignment)]
    lib.generate_consensus.restype = C.POINTER(_ConsensusData)
    lib.generate_consensus.argtypes = [C.POINTER(C.c_char_p), C.c_uint, C.c_uint, C.c_uint,
                                       C.c_double]
    lib.free_consensus_data.argtypes = [C.POINTER(_ConsensusData)]
    return lib


LEGACY_SYMBOLS = [
    "allocate_kmer_lookup", "init_kmer_lookup", "free_kmer_lookup", "allocate_seq",
    "init_seq_array", "free_seq_array", "allocate_seq_addr", "free_seq_addr_array",
    "add_sequence", "mask_k_mer", "find_kmer_pos_for_seq", "free_kmer_match",
    "find_best_aln_range", "find_best_aln_range2", "free_aln_range", "align",
    "free_alignment", "generate_consensus", "free_consensus_data",
]


class LegacyABI:
    """Drives any shared object exporting the reference's legacy C ABI through
    the same Python-level methods as ``Port`` (used for oracle/_ref and, in the
    GPU tests, for the product library itself)."""
    kind = "legacy-abi"

    def __init__(self, path: str):
        self.lib = bind_legacy_abi(C.CDLL(path))

    def find_hits(self, seed, query, K=8, mask=-1):
        lib = self.lib
        seed, query = _b(seed), _b(query)
        lk = lib.allocate_kmer_lookup(1 << (2 * K))
        sa = lib.allocate_seq(len(seed))
        sda = lib.allocate_seq_addr(len(seed))
        lib.add_sequence(0, K, seed, len(seed), sda, sa, lk)
        if mask >= 0:
            lib.mask_k_mer(1 << (2 * K), lk, mask)
        km = lib.find_kmer_pos_for_seq(query, len(query), K, sda, lk)
        n = km[0].count
        out = (list(km[0].query_pos[:n]), list(km[0].target_pos[:n]))
        lib.free_kmer_match(km)
        lib.free_seq_addr_array(sda)
        lib.free_seq_array(sa)
        lib.free_kmer_lookup(lk)
        return out

    def _range(self, fn, q, t, K, a, b):
        n = len(q)
        qa = (C.c_int * max(n, 1))(*q)
        ta = (C.c_int * max(n, 1))(*t)
        km = _KmerMatch(n, C.cast(qa, C.POINTER(C.c_int)), C.cast(ta, C.POINTER(C.c_int)))
        r = fn(C.byref(km), K, a, b)
        out = (r[0].s1, r[0].e1, r[0].s2, r[0].e2, r[0].score)
        self.lib.free_aln_range(r)
        return out

    def best_range(self, q, t, bin_size=48, count_th=5):
        if len(q) == 0:
            # the reference callocs a negative-sized histogram here
            # (kmer_lookup.c:344); harmless in C, but do not rely on it
            return (0, 0, 0, 0, 0)
        return self._range(self.lib.find_best_aln_range, q, t, 8, bin_size, count_th)

    def best_range2(self, q, t):
        if len(q) == 0:
            return (0, 0, 0, 0, 0)  # reference reads d_coor[0] of an empty array
        return self._range(self.lib.find_best_aln_range2, q, t, 8, 400, 25)

    def align(self, q, t, band=150, want_str=1):
        q, t = _b(q), _b(t)
        a = self.lib.align(q, len(q), t, len(t), band, want_str)
        r = a[0]
        out = dict(aln_str_size=r.aln_str_size, dist=r.dist, aln_q_s=r.aln_q_s,
                   aln_q_e=r.aln_q_e, aln_t_s=r.aln_t_s, aln_t_e=r.aln_t_e,
                   q_aln_str=C.string_at(r.q_aln_str).decode(),
                   t_aln_str=C.string_at(r.t_aln_str).decode())
        self.lib.free_alignment(a)
        return out

    def generate_consensus(self, seqs, min_cov=4, K=8, min_idt=0.70):
        seqs = [_b(s) for s in seqs]
        arr = (C.c_char_p * len(seqs))(*seqs)
        c = self.lib.generate_consensus(arr, len(seqs), min_cov, K, min_idt)
        seq = C.string_at(c[0].sequence).decode()
        eqv = list(c[0].eqv[:len(seq)])
        self.lib.free_consensus_data(c)
        return seq, eqv

    def generate_utg_consensus(self, seqs, offsets, min_cov=0, K=8, min_idt=0.70):
        """src/c/falcon.c:668-773 (exported, not bound by falcon_kit.py).  Returns the
        consensus, its eqv and the offsets array as the callee left it."""
        seqs = [_b(s) for s in seqs]
        arr = (C.c_char_p * len(seqs))(*seqs)
        off = (C.c_int * len(seqs))(*offsets)
        fn = self.lib.generate_utg_consensus
        fn.restype = C.POINTER(_ConsensusData)
        fn.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_uint, C.c_uint, C.c_uint,
                       C.c_double]
        c = fn(arr, off, len(seqs), min_cov, K, min_idt)
        seq = C.string_at(c[0].sequence).decode()
        eqv = list(c[0].eqv[:len(seq)])
        self.lib.free_consensus_data(c)
        return seq, eqv, list(off)


class Ref(LegacyABI):
    kind = "reference"

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(
                path + " is missing: run `make -C oracle` where /root/reference exists")
        super().__init__(path)


def have_ref() -> bool:
    return os.path.exists(REF_SO)
