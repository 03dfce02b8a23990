"""TEST INFRASTRUCTURE: python side of the SIMT emulator run of the consensus-stage kernels
(tests/emu/simt): takes piles (seed + reads as strings), produces what the alignment stage
hands the MSA stage -- windows, alignment summaries and edit scripts, here from the CPU oracle's
`find_hits` / `best_range` / `align` -- lays it out the way the engine lays a batch out in HBM,
runs the SOURCE of k_tags, k_tscan, k_links, k_score2 and k_backtrace on the host through
tests/emu/libemu_msa.so and returns the consensus strings, eqv arrays and per-pile score
records."""
import ctypes as C
import os
import subprocess

import numpy as np

from emu_driver import EMU_DIR, FaAln, FaPile, FaRange, FaSeq, pack

EMU_SO = os.path.join(EMU_DIR, "libemu_msa.so")
FaPileOut = np.dtype([("len", "<i4"), ("start", "<i4"), ("n_aligned", "<i4"), ("err", "<i4"), ("g_best_h", "<i8")])
FaScoreOut = np.dtype([("g_node", "<i4"), ("g_ck", "<i4"), ("g_h", "<i4"), ("n_levels", "<i4"),
                       ("n_links", "<i4"), ("err", "<i4"), ("wide", "<i4"), ("redo", "<i4")])
FaNode = np.dtype([("score_h", "<i4"), ("link", "<i4")])
assert FaPileOut.itemsize == 24 and FaScoreOut.itemsize == 32 and FaNode.itemsize == 8

_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", EMU_DIR, "libemu_msa.so"], check=True)
        _lib = C.CDLL(EMU_SO)
        _lib.emu_msa.restype = C.c_int
    return _lib


def script_of(q_aln, t_aln):
    """Edit script of a gapped alignment: one word per edit row, (snake << 1) | from_above;
    row 0 is the leading snake (the inverse of emu_driver.expand)."""
    rows = [0]
    for qc, tc in zip(q_aln, t_aln):
        if qc == "-":
            rows.append(1)          # target-only column: from above
        elif tc == "-":
            rows.append(0)          # query-only column
        else:
            rows[-1] += 2           # a match extends the row's snake
    return rows


def stage_piles(piles, port, min_idt=0.70, accept_all=False, band=150, whole=False, forced=None):
    """(arrays for emu_msa) from piles = [[seed, read, ...], ...] and an oracle Port."""
    seqs, words, woff = [], [], 0
    n_seq = sum(len(p) for p in piles)
    seq = np.zeros(n_seq, dtype=FaSeq)
    pile = np.zeros(len(piles), dtype=FaPile)
    rng = np.zeros(n_seq, dtype=FaRange)
    aln = np.zeros(n_seq, dtype=FaAln)
    script_off = np.zeros(n_seq, dtype=np.uint64)
    scripts = []
    so = 0
    g = 0
    for p, ss in enumerate(piles):
        seed = ss[0]
        pile[p]["first"] = g
        pile[p]["n_seq"] = len(ss)
        pile[p]["seed_len"] = len(seed)
        for j, s in enumerate(ss):
            w = pack(s)
            seq[g] = (woff, len(s), p, j)
            words.append(w)
            woff += len(w)
            script_off[g] = so
            if j > 0:
                hq, ht = port.find_hits(seed, s)
                s1, e1, s2, e2, score = port.best_range(hq, ht)
                if whole:  # (the windows of unitig consensus: whole sequences)
                    s1, e1, s2, e2 = 0, len(s), 0, len(seed)
                ok = not (e1 - s1 < 100 or e2 - s2 < 100 or
                          abs((e1 - s1) - (e2 - s2)) > int(0.5 * 0.10 * (e1 - s1 + e2 - s2)))  # falcon.c:613-619
                rng[g] = (s1, e1, s2, e2, 1 if ok else 0, len(hq), score)
                if forced and (p, j) in forced:  # a hand-made alignment of the whole read to the whole seed
                    qa, ta = forced[(p, j)]
                    assert qa.replace("-", "") == s and ta.replace("-", "") == seed and len(qa) == len(ta)
                    s1, e1, s2, e2, ok = 0, len(s), 0, len(seed), True
                    rng[g] = (s1, e1, s2, e2, 1, len(hq), score)
                if ok:
                    if forced and (p, j) in forced:
                        dist = sum(1 for x, y in zip(qa, ta) if x == "-" or y == "-")
                        a = dict(aln_str_size=len(qa), q_aln_str=qa, t_aln_str=ta, dist=dist, aln_q_e=len(s),
                                 aln_t_e=len(seed), cells=0)
                    else:
                        a = port.align(s[s1:e1], seed[s2:e2], band)
                    size = a["aln_str_size"]
                    rows = script_of(a["q_aln_str"], a["t_aln_str"])
                    assert len(rows) == a["dist"] + 1
                    accept = size > 500 and a["dist"] / size < 1.0 - min_idt   # falcon.c:629
                    n_ins = sum(1 for r in rows[1:] if not (r & 1))
                    aln[g] = (a["dist"], a["aln_q_e"], a["aln_t_e"], size, 1 if (accept or (accept_all and size > 0)) else 0,
                              n_ins, 1 if size > 0 else 0, 0, a["cells"])
                    scripts.append(np.asarray(rows, dtype=np.uint32))
                    so += (len(rows) + 3 + 3) & ~3
                    scripts.append(np.full(so - int(script_off[g]) - len(rows), 0xDEADBEEF, dtype=np.uint32))
            g += 1
    words = np.concatenate(words + [np.zeros(8, dtype=np.uint32)])
    script = np.concatenate(scripts + [np.full(8, 0xDEADBEEF, dtype=np.uint32)]) if scripts else np.zeros(8, np.uint32)
    return dict(words=words, seq=seq, pile=pile, rng=rng, aln=aln, script=script, script_off=script_off)


FaTInfo = np.dtype([("lvl_start", "<u4"), ("link_start", "<u4"), ("cov", "<u2"), ("nlev", "<u2")])


def run(st, min_cov=4, first_links_back=0, want_nodes=False, graph=None):
    """Run the staged batch; returns ([(consensus, eqv)], score_out records, nodes or None, pile records).
    graph: a dict that receives the position records, link words and links per level (debugging)."""
    pile = st["pile"].copy()
    n_pile, n_seq = len(pile), len(st["seq"])
    out_slots = int(sum(2 * int(t) + 4 for t in pile["seed_len"])) + 8
    out_seq = np.zeros(out_slots, dtype=np.uint8)
    out_eqv = np.zeros(out_slots, dtype=np.int32)
    pile_out = np.zeros(n_pile, dtype=FaPileOut)
    score_out = np.zeros(n_pile, dtype=FaScoreOut)
    node_cap = int(sum(int(t) + 2 for t in pile["seed_len"]) + st["aln"]["n_ins"][st["aln"]["accept"] != 0].sum()) * 5 + 8
    nodes = np.zeros(node_cap, dtype=FaNode) if want_nodes else None
    n_sync = C.c_ulonglong(0)
    acc = st["aln"]["accept"] != 0
    links_cap = int(st["aln"]["size"][acc].sum()) + 8 * n_pile + 8
    tinfo = np.zeros(int(pile["seed_len"].sum()) + 8, dtype=FaTInfo) if graph is not None else None
    links = np.zeros(links_cap, dtype=np.uint32) if graph is not None else None
    nlk = np.zeros(node_cap // 5 + 8, dtype=np.uint16) if graph is not None else None
    todo = np.zeros(6, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    rc = lib().emu_msa(p(st["words"]), C.c_uint64(len(st["words"])), p(st["seq"]), C.c_int(n_seq), p(pile), C.c_int(n_pile),
                       p(st["rng"]), p(st["aln"]), p(st["script"]), C.c_uint64(len(st["script"])), p(st["script_off"]),
                       C.c_uint(min_cov), C.c_int(first_links_back), p(out_seq), p(out_eqv), C.c_uint64(out_slots),
                       p(pile_out), p(score_out), p(nodes), C.c_uint64(node_cap), C.byref(n_sync),
                       p(tinfo), p(links), C.c_uint64(links_cap), p(nlk), p(todo))
    if graph is not None:
        graph.update(tinfo=tinfo, links=links, nlk=nlk, todo=[int(x) for x in todo])
    assert rc == 0, rc
    res = []
    for i in range(n_pile):
        o = int(pile[i]["out_off"]) + int(pile_out[i]["start"])
        n = int(pile_out[i]["len"])
        res.append((out_seq[o:o + n].tobytes().decode("ascii"), [int(x) for x in out_eqv[o:o + n]]))
    return res, score_out, nodes, pile
