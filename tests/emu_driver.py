"""TEST INFRASTRUCTURE: python side of the lane emulator (tests/emu): lays (query window,
target window) pairs out the way the engine lays a batch out in HBM (2-bit packed words,
FaSeq / FaPile / FaRange records, the work queue), runs the k_align2 kernel source on the
host through tests/emu/libemu_align2.so and returns the alignment summaries and the gapped
strings expanded from the edit scripts -- the shape oracle.pyoracle.Port.align() returns."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
EMU_SO = os.path.join(EMU_DIR, "libemu_align2.so")

FaSeq = np.dtype([("woff", "<u4"), ("len", "<i4"), ("pile", "<i4"), ("idx", "<i4")])
FaPile = np.dtype([("first", "<i4"), ("n_seq", "<i4"), ("seed_len", "<i4"), ("pad0", "<i4"),
                   ("kidx_off", "<u8"), ("kpos_off", "<u8"), ("node_off", "<u8"), ("node_cap", "<u8"),
                   ("out_off", "<u8")])
FaRange = np.dtype([("s1", "<i4"), ("e1", "<i4"), ("s2", "<i4"), ("e2", "<i4"), ("ok", "<i4"),
                    ("n_hit", "<i4"), ("score", "<i8")])
FaAln = np.dtype([("dist", "<i4"), ("q_e", "<i4"), ("t_e", "<i4"), ("size", "<i4"), ("accept", "<i4"),
                  ("n_ins", "<i4"), ("aligned", "<i4"), ("err", "<i4"), ("cells", "<i8")])
assert FaSeq.itemsize == 16 and FaPile.itemsize == 56 and FaRange.itemsize == 32 and FaAln.itemsize == 40

_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def build():
    subprocess.run(["make", "-s", "-C", EMU_DIR], check=True)
    return EMU_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(EMU_SO)
        _lib.emu_align2.restype = C.c_int
    return _lib


def pack(seq: str) -> np.ndarray:
    """2 bits per base, 16 per u32, base i at bits 2 (i mod 16); >= 2 zero words behind."""
    codes = _CODE[np.frombuffer(seq.encode("ascii"), dtype=np.uint8)]
    assert (codes < 4).all()
    nw = (len(seq) + 15) // 16 + 2
    nw = (nw + 3) & ~3
    padded = np.zeros(nw * 16, dtype=np.uint32)
    padded[:len(seq)] = codes
    sh = (np.arange(16, dtype=np.uint32) * 2)[None, :]
    return (padded.reshape(nw, 16) << sh).sum(axis=1, dtype=np.uint64).astype(np.uint32)


def expand(script, dist, q, t):
    """Gapped strings from an edit script ((snake << 1) | from_above per row)."""
    qs, ts = [], []
    x = y = 0
    for d in range(dist + 1):
        e = int(script[d])
        if d > 0:
            if e & 1:
                qs.append("-"); ts.append(t[y]); y += 1
            else:
                qs.append(q[x]); ts.append("-"); x += 1
        m = e >> 1
        qs.append(q[x:x + m]); ts.append(t[y:y + m])
        x += m; y += m
    return "".join(qs), "".join(ts), x, y


def align_pairs(pairs, band=150, ring=8192, order=None, max_diff=2.0, windows=None, n_wave=1):
    """pairs: [(query, target)].  windows: optional [(s1, e1, s2, e2)] per pair (default: the
    whole strings).  Returns ([result dict per pair], stats[12]); a result has the keys of
    Port.align() plus `err`, `n_ins` and `accept`."""
    n = len(pairs)
    seqs, words, woff = [], [], 0
    for q, t in pairs:
        for s in (t, q):  # pile = (target, query), like fa_align_pairs
            w = pack(s)
            seqs.append((woff, len(s)))
            words.append(w)
            woff += len(w)
    words = np.concatenate(words + [np.zeros(8, dtype=np.uint32)])
    seq = np.zeros(2 * n, dtype=FaSeq)
    pile = np.zeros(n, dtype=FaPile)
    rng = np.zeros(2 * n, dtype=FaRange)
    script_off = np.zeros(2 * n, dtype=np.uint64)
    so = 0
    for i, (q, t) in enumerate(pairs):
        for j in (0, 1):
            g = 2 * i + j
            seq[g] = (seqs[g][0], seqs[g][1], i, j)
        pile[i]["first"] = 2 * i
        pile[i]["n_seq"] = 2
        pile[i]["seed_len"] = len(t)
        s1, e1, s2, e2 = windows[i] if windows else (0, len(q), 0, len(t))
        rng[2 * i + 1] = (s1, e1, s2, e2, 1, 0, 0)
        script_off[2 * i + 1] = so
        so += (int(0.3 * ((e1 - s1) + (e2 - s2))) + 2 + 3) & ~3
    script = np.full(so + 8, 0xDEADBEEF, dtype=np.uint32)
    aln = np.zeros(2 * n, dtype=FaAln)
    aln["err"] = -77  # (every record must be written)
    if order is None:
        order = np.arange(2 * n, dtype=np.int32)
    order = np.ascontiguousarray(order, dtype=np.int32)
    in_queue = set(int(g) for g in order)
    stats = np.zeros(12, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().emu_align2(p(words), C.c_uint64(len(words)), p(seq), C.c_int(2 * n), p(pile), C.c_int(n),
                          p(rng), p(order), C.c_int(len(order)), C.c_uint32(ring), C.c_int(n_wave),
                          C.c_int(band), C.c_double(max_diff), p(script), C.c_uint64(len(script)),
                          p(script_off), p(aln), p(stats))
    assert rc == 0
    out = []
    for i, (q, t) in enumerate(pairs):
        a = aln[2 * i + 1]
        assert aln[2 * i]["err"] == 0 or (2 * i) not in in_queue, "target's record not written"
        assert (2 * i + 1) in in_queue, "the pair was not in the work queue"
        s1, e1, s2, e2 = windows[i] if windows else (0, len(q), 0, len(t))
        r = dict(err=int(a["err"]), aligned=int(a["aligned"]), dist=int(a["dist"]), aln_q_e=int(a["q_e"]),
                 aln_t_e=int(a["t_e"]), aln_str_size=int(a["size"]), cells=int(a["cells"]),
                 n_ins=int(a["n_ins"]), accept=int(a["accept"]), aln_q_s=0, aln_t_s=0, q_aln_str="",
                 t_aln_str="")
        if r["aligned"]:
            sc = script[int(script_off[2 * i + 1]):]
            try:
                qs, ts, x, y = expand(sc, r["dist"], q[s1:e1], t[s2:e2])
            except IndexError:
                raise AssertionError("pair %d: the edit script runs past the sequences: %r" % (i, r))
            assert (x, y) == (r["aln_q_e"], r["aln_t_e"]), (i, x, y, r)
            r["q_aln_str"], r["t_aln_str"] = qs, ts
            r["aln_str_size"] = len(qs)
            r["n_ins_script"] = sum(1 for d in range(1, r["dist"] + 1) if not (int(sc[d]) & 1))
        out.append(r)
    return out, stats
