"""The seed-index kernel's SOURCE (falcon_amd/csrc/k_seed_index.hip: the 8-mer table in the LDS) on the
host-side SIMT emulator of tests/emu/simt, against a numpy statement of what the reference's lookup
enumerates: for every 8-mer the seed positions 0 .. len-9 that start it, ascending
(src/c/kmer_lookup.c:140-192, :174).  No GPU."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
NKMER = 65536
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", EMU_DIR, "libemu_index.so"], check=True)
        _lib = C.CDLL(os.path.join(EMU_DIR, "libemu_index.so"))
        _lib.emu_seed_index.restype = C.c_longlong
        _lib.emu_seed_index.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    return _lib


def pack(codes):
    """2 bits per base, 16 per u32, base i at bits 2*(i % 16); two zero words behind (fa_internal.h)."""
    n = len(codes)
    padded = np.zeros((n + 15) // 16 * 16, dtype=np.uint32)
    padded[:n] = codes
    words = (padded.reshape(-1, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint64).astype(np.uint32)
    return np.concatenate([words, np.zeros(2, dtype=np.uint32)])


def expected(codes):
    n_pos = max(0, len(codes) - 8)  # kmer_lookup.c:174: the last 8-mer is not indexed
    if n_pos == 0:
        return np.zeros(NKMER + 1, dtype=np.uint32), np.zeros(0, dtype=np.uint32)
    c = np.asarray(codes, dtype=np.uint32)
    k = np.zeros(n_pos, dtype=np.uint32)
    for j in range(8):
        k |= c[j:j + n_pos] << (2 * j)
    P = np.argsort(k, kind="stable").astype(np.uint32)
    T = np.zeros(NKMER + 1, dtype=np.uint32)
    T[1:] = np.cumsum(np.bincount(k, minlength=NKMER))
    return T, P


def run(codes, use_long=0):
    codes = np.asarray(codes, dtype=np.uint32)
    words = pack(codes)
    T = np.zeros(NKMER + 1, dtype=np.uint32)
    P = np.zeros(max(1, len(codes)), dtype=np.uint32)
    n = lib().emu_seed_index(words.ctypes.data, len(codes), T.ctypes.data, P.ctypes.data, use_long)
    assert n >= 0
    return T, P[:max(0, len(codes) - 8)]


def check(codes, use_long=0):
    T, P = run(codes, use_long)
    eT, eP = expected(codes)
    assert np.array_equal(T, eT)
    assert np.array_equal(P, eP)


def test_random_seeds_of_bench_lengths():
    rng = np.random.default_rng(5)
    for n in (20000, 30011, 12345):
        check(rng.integers(0, 4, n))


def test_short_and_empty_seeds():
    rng = np.random.default_rng(6)
    for n in (0, 1, 7, 8, 9, 10, 63, 64, 65, 71, 72, 73, 255, 256 + 8, 257 + 8, 1023 + 8, 1024 + 8, 1025 + 8):
        check(rng.integers(0, 4, n))


def test_repeats_put_one_8mer_many_times_into_a_step():
    """Homopolymers, tandem repeats of period 1-9 and a seed of only two 8-mers: steps of the fill in
    which an 8-mer occurs 2 .. 64 times (the ballot ranking), buckets of thousands."""
    rng = random.Random(7)
    check([0] * 5000)
    check([3] * 777)
    for period in (2, 3, 5, 7, 9, 31, 64, 65):
        unit = [rng.randrange(4) for _ in range(period)]
        check((unit * (4000 // period + 2))[:4000])
    # random sequence with repeat islands
    seq = [rng.randrange(4) for _ in range(9000)]
    for _ in range(12):
        at = rng.randrange(0, 8000)
        unit = [rng.randrange(4) for _ in range(rng.choice([1, 2, 3, 4, 6]))]
        n = rng.randrange(20, 400)
        seq[at:at + n] = (unit * (n // len(unit) + 1))[:n]
    check(seq)


def test_the_longest_seed_the_lds_table_takes():
    """65 543 bases = 65 535 positions: every cursor still fits 16 bits -- and a homopolymer of that
    length puts all of them into one bucket."""
    rng = np.random.default_rng(8)
    check(rng.integers(0, 4, 65543))
    check(np.zeros(65543, dtype=np.uint32))
    check(np.full(65543, 3, dtype=np.uint32))
    check(np.zeros(65544, dtype=np.uint32))  # one more: the kernel behind takes it
    check(rng.integers(0, 4, 70001))


def test_the_kernel_behind_on_short_seeds_too():
    """k_seed_index_long (the table in global memory: rounds 1-3's kernel) with every_pile set, as
    FALCON_AMD_INDEX_LONG runs it."""
    rng = np.random.default_rng(9)
    check(rng.integers(0, 4, 5000), use_long=1)
    check([1] * 900, use_long=1)
    check(rng.integers(0, 4, 7), use_long=1)
