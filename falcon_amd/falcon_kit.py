"""ctypes face of libfalcon_amd.so with the names the reference package exposes
(/root/reference/falcon_kit/falcon_kit.py:1-122): ``kup``, ``DWA`` and ``falcon``
are one CDLL; the Structure classes mirror src/c/common.h:59-126.

A reference user does ``from falcon_kit import kup, DWA, falcon`` -- pointing that
import at this module swaps in the GPU implementation without
touching any caller.  ``generate_consensus`` is bound with its true 5-argument C
signature (src/c/falcon.c:562-566; the reference module's 7-argument declaration at
falcon_kit.py:119-120 is stale and is overridden by consensus.py:20-21)."""
from __future__ import annotations

from ctypes import (POINTER, Structure, c_char_p, c_double, c_int, c_long, c_uint, c_uint8,
                    c_void_p, string_at)

from .lib import load

__all__ = ["kup", "DWA", "falcon", "KmerLookup", "KmerMatch", "AlnRange", "ConsensusData",
           "Alignment", "get_alignment"]

seq_coor_t = c_int
base_t = c_uint8


class KmerLookup(Structure):          # common.h:95-99
    _fields_ = [("start", seq_coor_t), ("last", seq_coor_t), ("count", seq_coor_t)]


class KmerMatch(Structure):           # common.h:107-111
    _fields_ = [("count", seq_coor_t), ("query_pos", POINTER(seq_coor_t)),
                ("target_pos", POINTER(seq_coor_t))]


class AlnRange(Structure):            # common.h:114-120
    _fields_ = [("s1", seq_coor_t), ("e1", seq_coor_t), ("s2", seq_coor_t),
                ("e2", seq_coor_t), ("score", c_long)]


class ConsensusData(Structure):       # common.h:123-126
    _fields_ = [("sequence", c_void_p), ("eff_cov", POINTER(c_uint))]


class Alignment(Structure):           # common.h:59-69
    _fields_ = [("aln_str_size", seq_coor_t), ("dist", seq_coor_t), ("aln_q_s", seq_coor_t),
                ("aln_q_e", seq_coor_t), ("aln_t_s", seq_coor_t), ("aln_t_e", seq_coor_t),
                ("q_aln_str", c_void_p), ("t_aln_str", c_void_p)]


_dll = load()
kup = DWA = falcon = _dll

_dll.allocate_kmer_lookup.argtypes = [seq_coor_t]
_dll.allocate_kmer_lookup.restype = POINTER(KmerLookup)
_dll.init_kmer_lookup.argtypes = [POINTER(KmerLookup), seq_coor_t]
_dll.free_kmer_lookup.argtypes = [POINTER(KmerLookup)]
_dll.allocate_seq.argtypes = [seq_coor_t]
_dll.allocate_seq.restype = POINTER(base_t)
_dll.init_seq_array.argtypes = [POINTER(base_t), seq_coor_t]
_dll.free_seq_array.argtypes = [POINTER(base_t)]
_dll.allocate_seq_addr.argtypes = [seq_coor_t]
_dll.allocate_seq_addr.restype = POINTER(seq_coor_t)
_dll.free_seq_addr_array.argtypes = [POINTER(seq_coor_t)]
_dll.add_sequence.argtypes = [seq_coor_t, c_uint, c_char_p, seq_coor_t, POINTER(seq_coor_t),
                              POINTER(base_t), POINTER(KmerLookup)]
_dll.mask_k_mer.argtypes = [seq_coor_t, POINTER(KmerLookup), seq_coor_t]
_dll.find_kmer_pos_for_seq.argtypes = [c_char_p, seq_coor_t, c_uint, POINTER(seq_coor_t),
                                       POINTER(KmerLookup)]
_dll.find_kmer_pos_for_seq.restype = POINTER(KmerMatch)
_dll.free_kmer_match.argtypes = [POINTER(KmerMatch)]
for _f in (_dll.find_best_aln_range, _dll.find_best_aln_range2):
    _f.argtypes = [POINTER(KmerMatch), seq_coor_t, seq_coor_t, seq_coor_t]
    _f.restype = POINTER(AlnRange)
_dll.free_aln_range.argtypes = [POINTER(AlnRange)]
_dll.align.argtypes = [c_char_p, seq_coor_t, c_char_p, seq_coor_t, seq_coor_t, c_int]
_dll.align.restype = POINTER(Alignment)
_dll.free_alignment.argtypes = [POINTER(Alignment)]
_dll.generate_consensus.argtypes = [POINTER(c_char_p), c_uint, c_uint, c_uint, c_double]
_dll.generate_consensus.restype = POINTER(ConsensusData)
_dll.free_consensus_data.argtypes = [POINTER(ConsensusData)]


def consensus_of(seqs, min_cov=6, K=8, min_idt=0.70):
    """One pile through the legacy entry point (a GPU batch of one)."""
    raw = [s if isinstance(s, bytes) else s.encode("ascii") for s in seqs]
    arr = (c_char_p * len(raw))(*raw)
    ptr = _dll.generate_consensus(arr, len(raw), min_cov, K, min_idt)
    try:
        return string_at(ptr[0].sequence).decode("ascii")
    finally:
        _dll.free_consensus_data(ptr)


def get_alignment(seq1, seq0):
    """The helper of falcon_kit.py:125-179: chain seq1 on seq0 with the k-mer table
    functions, extend the window by 2K, align it with band 100 and report
    (q_start, q_end, t_start, t_end, columns, distance) or None."""
    K = 8
    b1 = seq1 if isinstance(seq1, bytes) else seq1.encode("ascii")
    b0 = seq0 if isinstance(seq0, bytes) else seq0.encode("ascii")
    table = kup.allocate_kmer_lookup(1 << (2 * K))
    codes = kup.allocate_seq(len(b0))
    chain = kup.allocate_seq_addr(len(b0))
    kup.add_sequence(0, K, b0, len(b0), chain, codes, table)
    hits = kup.find_kmer_pos_for_seq(b1, len(b1), K, chain, table)
    rng = kup.find_best_aln_range(hits, K, K * 10, 50)
    s1, e1, s2, e2 = rng[0].s1, rng[0].e1, rng[0].s2, rng[0].e2
    kup.free_kmer_match(hits)
    kup.free_aln_range(rng)
    kup.free_seq_addr_array(chain)
    kup.free_seq_array(codes)
    kup.free_kmer_lookup(table)
    if e1 - s1 <= 500:
        return None
    e1 = len(b1) if e1 >= len(b1) - 2 * K else e1 + 2 * K
    e2 = len(b0) if e2 >= len(b0) - 2 * K else e2 + 2 * K
    aln = DWA.align(b1[s1:e1], e1 - s1, b0[s2:e2], e2 - s2, 100, 0)
    size, dist = aln[0].aln_str_size, aln[0].dist
    q_e, t_e = aln[0].aln_q_e - aln[0].aln_q_s, aln[0].aln_t_e - aln[0].aln_t_s
    DWA.free_alignment(aln)
    if size <= 500:
        return None
    return s1, s1 + q_e, s2, s2 + t_e, size, dist
