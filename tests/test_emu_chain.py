"""The chaining kernel's SOURCE (falcon_amd/csrc/k_chain.hip: k-mer hits -> best co-linear window) on
the host-side SIMT emulator of tests/emu/simt, against the CPU oracle's find_kmer_pos_for_seq /
find_best_aln_range (src/c/kmer_lookup.c:207-427) and the range filter of falcon.c:613-619; the
per-probe records the kernel's first pass leaves are decoded back into the hit list.  No GPU."""
import ctypes as C
import os
import random
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from test_emu_index import expected as index_of, pack  # noqa: E402
from oracle.pyoracle import Port  # noqa: E402
from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs  # noqa: E402

EMU_DIR = os.path.join(HERE, "emu")
_lib = None


class FaRange(C.Structure):
    _fields_ = [("s1", C.c_int), ("e1", C.c_int), ("s2", C.c_int), ("e2", C.c_int), ("ok", C.c_int),
                ("n_hit", C.c_int), ("score", C.c_longlong)]


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", EMU_DIR, "libemu_chain.so"], check=True)
        _lib = C.CDLL(os.path.join(EMU_DIR, "libemu_chain.so"))
        _lib.emu_chain.restype = C.c_longlong
    return _lib


@pytest.fixture(scope="module")
def port():
    return Port()


CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def run_pile(pile, lds_bins=None):
    """pile: [seed, read, ...] as strings -> ([FaRange per sequence], [hit list per sequence])"""
    codes = [np.array([CODE[c] for c in s], dtype=np.uint32) for s in pile]
    words, woff = [], []
    at = 0
    for c in codes:
        w = pack(c)
        w = np.concatenate([w, np.zeros((-len(w)) % 4, dtype=np.uint32)])  # 16-byte aligned starts
        woff.append(at)
        words.append(w)
        at += len(w)
    words = np.concatenate(words + [np.zeros(4, dtype=np.uint32)])
    lens = np.array([len(c) for c in codes], dtype=np.int32)
    T, P = index_of(codes[0])
    n_probe = [((n - 8 + 3) // 4 if n > 8 else 0) for n in lens]
    poff = np.concatenate([[0], np.cumsum(n_probe)]).astype(np.uint64)
    if lds_bins is None:
        lds_bins = (int(lens.max()) + int(lens[0])) // 48 + 8
    out = (FaRange * len(pile))()
    rec = np.zeros(int(poff[-1]) + 1, dtype=np.uint64)
    woff = np.array(woff, dtype=np.uint64)
    Pbuf = np.concatenate([P, np.zeros(1, dtype=np.uint32)])
    lib().emu_chain(words.ctypes.data_as(C.c_void_p), C.c_longlong(len(words)), woff.ctypes.data_as(C.c_void_p),
                    lens.ctypes.data_as(C.c_void_p), C.c_int(len(pile)), T.ctypes.data_as(C.c_void_p),
                    Pbuf.ctypes.data_as(C.c_void_p), C.c_int(lds_bins), out, rec.ctypes.data_as(C.c_void_p),
                    poff.ctypes.data_as(C.c_void_p), C.c_longlong(int(poff[-1])))
    hits = []
    for g in range(len(pile)):  # the records, decoded the way fa_batch_debug_hits does
        hq, ht = [], []
        for p in range(n_probe[g]):
            r = int(rec[int(poff[g]) + p])
            n, a, b = r & 3, (r >> 2) & 0x1ffff, (r >> 19) & 0x1ffff
            ts = [] if n == 0 else [a] if n == 1 else [a, b] if n == 2 else [int(x) for x in P[a:a + b]]
            hq += [4 * p] * len(ts)
            ht += ts
        hits.append((hq, ht))
    return list(out), hits


def check_pile(pile, port, lds_bins=None):
    got, hits = run_pile(pile, lds_bins)
    assert (got[0].s1, got[0].e1, got[0].s2, got[0].e2, got[0].ok, got[0].n_hit) == (0, 0, 0, 0, 0, 0)
    n_ok = 0
    for g in range(1, len(pile)):
        hq, ht = port.find_hits(pile[0], pile[g])
        r = got[g]
        assert r.n_hit == len(hq), g
        if len(hq) == 0:
            assert (r.s1, r.e1, r.s2, r.e2, r.ok) == (0, 0, 0, 0, 0)
            continue
        assert hits[g] == (hq, ht), g
        assert (r.s1, r.e1, r.s2, r.e2, r.score) == tuple(port.best_range(hq, ht)), g
        dq, dt = r.e1 - r.s1, r.e2 - r.s2
        want_ok = not (dq < 100 or dt < 100 or abs(dq - dt) > int(0.5 * 0.10 * (dq + dt)))  # falcon.c:613-619
        assert r.ok == int(want_ok), g
        n_ok += r.ok
    return n_ok


def test_synthetic_piles(port):
    n_ok = 0
    for seed, kw in ((41, dict(S=5000, coverage=15, min_read=800, mean_read=3000, sd_read=1000)),
                     (42, dict(S=9000, coverage=12, e=0.20)),
                     (43, dict(S=4000, coverage=10, e=0.02, min_read=2500, mean_read=3500, sd_read=300)),
                     (44, dict(S=7000, coverage=10, het=0.01))):
        s, rd = make_pile(seed, **kw)
        pile = [codes_to_str(x) for x in pile_to_seqs(s, rd)]
        n_ok += check_pile(pile, port)
    assert n_ok > 40


def test_repeats_low_complexity_and_unrelated_reads(port):
    """Buckets of 3, 4, 5 .. hundreds of entries (the records' third form, the loop beyond four
    entries), reads that hit nothing or only by chance, reads shorter than a k-mer."""
    rng = random.Random(9)

    def rnd(n):
        return "".join(rng.choice("ACGT") for _ in range(n))

    def noisy(s, e):
        out = []
        for c in s:
            x = rng.random()
            if x < e / 3:
                continue
            if x < 2 * e / 3:
                out.append(rng.choice("ACGT"))
            if x < e:
                out.append(rng.choice("ACGT"))
                continue
            out.append(c)
        return "".join(out)
    unit = rnd(300)
    seed = rnd(1500) + unit * 4 + rnd(800) + "ACG" * 150 + rnd(1200) + "A" * 300 + rnd(900) + unit + rnd(400)
    reads = [noisy(seed[a:a + n], e) for a, n, e in ((0, 4000, 0.1), (1000, 3500, 0.12), (2500, 3000, 0.05),
                                                     (1400, 1300, 0.08), (3400, 1500, 0.1), (4800, 1800, 0.1),
                                                     (0, len(seed), 0.15), (200, 900, 0.0))]
    reads += [rnd(2500), rnd(30), "ACGTACG", "ACGTACGT", "ACGTACGTA", "A" * 400, "ACG" * 200, unit * 3]
    check_pile([seed, seed] + reads, port)


def test_too_many_bins_for_the_lds_is_reported(port):
    s, rd = make_pile(45, S=3000, coverage=6)
    pile = [codes_to_str(x) for x in pile_to_seqs(s, rd)]
    got, _ = run_pile(pile, lds_bins=4)
    assert any(r.ok == -1 for r in got[1:])
