"""The consensus-stage kernels' SOURCE (k_msa.hip: k_tags, k_sscan, k_links, k_backtrace;
k_links2.hip: k_links2; k_score2.hip: k_score2, a workgroup of two wavefronts) on the host-side SIMT
emulator of tests/emu/simt, against the CPU oracle
and, stage by stage, against the plain-python graph model of tests/msa_model.py.  No GPU: this is
what pins the kernels' logic here, in the dev container; the GPU suite then only has to confirm
that the hardware runs the same source the same way.  (src/c/falcon.c:106-162, :232-263,
:308-558)"""
import os
import random
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import emu_msa_driver as D  # noqa: E402
from emu_driver import expand  # noqa: E402
from msa_model import Graph, tags_of  # noqa: E402
from oracle.campaign_cases import pile_cases  # noqa: E402
from oracle.pyoracle import Port  # noqa: E402
from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs  # noqa: E402

TSEG = 128  # k_msa.h


@pytest.fixture(scope="module")
def port():
    return Port()


def model_of(st, pile):
    """The model's graph from the staged alignments (their edit scripts expanded again)."""
    G = Graph(len(pile[0]))
    for g in range(1, len(pile)):
        a = st["aln"][g]
        if not a["accept"]:
            continue
        r = st["rng"][g]
        sc = st["script"][int(st["script_off"][g]):]
        qs, ts, _x, _y = expand(sc, int(a["dist"]), pile[g][r["s1"]:r["e1"]], pile[0][r["s2"]:r["e2"]])
        G.add(tags_of(qs, ts, int(r["s1"]), int(r["s2"])))
    return G


def check_stages(st, pile, min_cov, todo=None, scores=True):
    """One pile through the emulated kernels; every intermediate product against the model.
    todo (a list) receives the number of segments each k_links instance was handed."""
    graph = {}
    res, so, nodes, _pl = D.run(st, min_cov, want_nodes=True, graph=graph)
    if todo is not None:
        todo[:] = graph["todo"]
    G = model_of(st, pile)
    ls, ks, nlev, n_lvl, n_lnk = G.layout()
    ti = graph["tinfo"]
    for t in range(G.T):  # k_tags' segment sums + k_sscan + k_links2's position records
        assert (int(ti[t]["lvl_start"]), int(ti[t]["link_start"]), int(ti[t]["cov"]), int(ti[t]["nlev"])) == \
            (ls[t], ks[t], min(G.cov[t], 65535), nlev[t]), t
    assert (int(so[0]["n_levels"]), int(so[0]["n_links"])) == (n_lvl, n_lnk)
    k = 0
    for t in range(G.T):  # k_links: a segment's links back to back from its first link slot
        if t % TSEG == 0:
            k = ks[t]
        if G.cov[t] == 0:
            continue
        for d in range(nlev[t]):
            exp = G.link_words(t, d, ls)
            assert [int(x) for x in graph["links"][k:k + len(exp)]] == exp, (t, d)
            assert int(graph["nlk"][ls[t] + d]) == len(exp), (t, d)
            k += len(exp)
    if not scores:  # (positions with more levels than k_score2 takes: the graph is what is checked)
        return res[0]
    assert so[0]["redo"] == 0 and so[0]["err"] == 0
    sc, best = G.scores(ls)  # k_score2
    t_of_slot = {ls[t] + d: t for t in range(G.T) for d in range(nlev[t])}
    for nid, (h, bp, _bk) in sc.items():
        got = nodes[nid]
        assert int(got["score_h"]) == h, (nid, got, h)
        assert int(got["link"]) >> 1 == (bp if bp is not None else 0) + 1, (nid, got, bp)
        assert int(got["link"]) & 1 == (1 if G.cov[t_of_slot[nid // 5]] > min_cov else 0), nid  # falcon.c:498
    if best[1] >= 0:
        assert (int(so[0]["g_h"]), int(so[0]["g_node"]), int(so[0]["g_ck"])) == best
    else:
        assert int(so[0]["g_node"]) == -1
    return res[0]


def test_smoke_pile_every_stage(port):
    s, rd = make_pile(5, S=4000, coverage=14, min_read=800, mean_read=2500, sd_read=800)
    pile = [codes_to_str(x) for x in pile_to_seqs(s, rd)]
    st = D.stage_piles([pile], port)
    todo = []
    got = check_stages(st, pile, 4, todo)
    assert got == tuple(port.generate_consensus(pile, 4, 8, 0.70))
    assert todo == [0, 0, 0, 0, 0, 0]  # k_links2 held every segment


def test_k_links_takes_every_segment(port, monkeypatch):
    """The fallback kernels (k_links<1..16>, lanes = alignments) on their own: k_links2 hands them
    every segment (FALCON_AMD_LINKS1 in the product)."""
    monkeypatch.setenv("EMU_MSA_LINKS1", "1")
    for i, (pile, mc, idt) in enumerate(pile_cases(5)[:6]):
        st = D.stage_piles([pile], port, min_idt=idt)
        todo = []
        got = check_stages(st, pile, mc, todo)
        assert todo[0] == todo[1] == (len(pile[0]) + TSEG - 1) // TSEG
        assert got == tuple(port.generate_consensus(pile, mc, 8, idt)), i


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_campaign_piles_vs_oracle(port, seed):
    """Twelve piles per seed of the differential campaign's generator: depth 4-60x, error 1-25 %,
    unrelated reads, low-complexity seeds, min_cov 0-8, min_idt 0.60-0.95; consensus and eqv
    against the oracle, four of them stage by stage against the model as well."""
    cases = pile_cases(seed)
    sts = [D.stage_piles([p], port, min_idt=idt) for (p, _mc, idt) in cases]
    for i, ((pile, mc, idt), st) in enumerate(zip(cases, sts)):
        exp = tuple(port.generate_consensus(pile, mc, 8, idt))
        if i % 3 == 0:
            got = check_stages(st, pile, mc)
        else:
            res, so, _n, _p = D.run(st, mc)
            assert so[0]["redo"] == 0
            got = res[0]
        assert got == exp, (seed, i, len(got[0]), len(exp[0]))


def test_a_batch_of_piles_in_one_launch(port):
    """Several piles in one batch: per-pile offsets into every pool, one wavefront per pile."""
    cases = pile_cases(7)[:5]
    piles = [c[0] for c in cases]
    st = D.stage_piles(piles, port, min_idt=0.70)
    res, so, _n, _p = D.run(st, 2)
    for pile, got in zip(piles, res):
        assert got == tuple(port.generate_consensus(pile, 2, 8, 0.70))


def test_deep_pile_many_links_per_level(port):
    """A deep, noisy pile: levels with more than 16 links take the 64-lane prefix maximum of
    k_score2's chain (365 accepted alignments on a 3 kb seed)."""
    s, rd = make_pile(77, S=3000, coverage=250, e=0.13, min_read=1500, mean_read=2500, sd_read=500)
    pile = [codes_to_str(x) for x in pile_to_seqs(s, rd, 600)]
    st = D.stage_piles([pile], port)
    graph = {}
    res, so, _n, _p = D.run(st, 4, graph=graph)
    assert so[0]["redo"] == 0
    assert int(graph["nlk"][:int(so[0]["n_levels"])].max()) > 16
    assert res[0] == tuple(port.generate_consensus(pile, 4, 8, 0.70))


@pytest.mark.parametrize("clustered", [False, True])
def test_long_insertion_runs(port, clustered):
    """Reads carrying runs of 13-40 inserted bases: tag words with out-of-line runs, positions
    with dozens of levels, blocks of k_score2 that hold a single position.  Scattered, k_links2's
    pool of listed groups holds them; with every read's runs inside the same 60 bases the pool
    of those 64 positions overflows and k_links takes that segment -- and only that one."""
    rng = random.Random(11)
    s, rd = make_pile(31, S=3000, coverage=12, e=0.08, min_read=1200, mean_read=2200, sd_read=400)
    seed = codes_to_str(s)
    reads = []
    for r in rd:
        r = list(codes_to_str(r))
        for _ in range(3):
            at = rng.randrange(200, len(r) - 200)
            r[at:at] = [rng.choice("ACGT") for _ in range(rng.choice([13, 20, 40]))]
        reads.append("".join(r))
    if clustered:  # fourteen copies of the seed, each with three runs somewhere in seed[1500:1540]
        for _ in range(14):
            r = list(seed)
            for _k in range(3):
                at = 1500 + rng.randrange(0, 40)
                r[at:at] = [rng.choice("ACGT") for _ in range(rng.choice([13, 20, 40]))]
            for _k in range(30):
                del r[rng.randrange(50, len(r) - 50)]
            reads.append("".join(r))
    pile = [seed, seed] + reads
    st = D.stage_piles([pile], port)
    todo = []
    got = check_stages(st, pile, 2, todo)
    assert got == tuple(port.generate_consensus(pile, 2, 8, 0.70))
    n_seg = (len(seed) + TSEG - 1) // TSEG
    # (the instance with the large pool holds what the first one hands on)
    # (scattered, one segment of the 24 may still meet three runs in its 64 positions: 128 listed groups)
    assert (0 < todo[0] < n_seg // 2 and todo[1] == 0) if clustered else (todo[0] <= 1 and todo[1] == 0), todo


def test_pile_deeper_than_1023_alignments(port):
    """~1200 accepted alignments on a 2.5 kb seed (--max-n-read 2000): k_links2 walks them 64 at
    a time, link counts beyond 10 bits, coverage beyond 1023 in k_score2's biased scores
    (rounds 1-3 reported such piles and left them uncorrected; falcon.c:597-647 loops over any
    n_seq)."""
    s, rd = make_pile(712, S=2500, coverage=830, e=0.08, min_read=1500, mean_read=2200, sd_read=200)
    pile = [codes_to_str(x) for x in pile_to_seqs(s, rd, 5000)]
    st = D.stage_piles([pile], port)
    assert int(st["aln"]["accept"].sum()) > 1100
    graph = {}
    res, so, _n, _p = D.run(st, 4, graph=graph)
    assert so[0]["redo"] == 0 and so[0]["err"] == 0
    assert graph["todo"][0] > 0 and graph["todo"][1] == 0  # the large pool's instance took segments, k_links none
    assert int(graph["links"].max() & 0xffff) > 1023
    assert res[0] == tuple(port.generate_consensus(pile, 4, 8, 0.70))


@pytest.mark.parametrize("runs", [[20, 40, 120, 200], [17, 300, 260], [254, 255, 256], [400], [64, 128, 192, 253]])
def test_insertion_runs_past_the_255_column_cut_off(port, runs):
    """Alignments (hand-made: a copy of the seed with one block of inserted bases, as the wide bands of
    unitig consensus and contig layout can produce -- falcon_sense's band of 150 never gets there) with
    runs of up to 400 inserted bases: tagging stops at the first column whose insertion depth reaches
    255 (falcon.c:138-152), the alignment covers only what came before it.  k_tags no longer walks
    every script twice to find that column: its main pass looks for a chunk of 64 all-zero script
    words and only then finds the exact row.  Position records and link words against the model
    (tests/msa_model.py breaks at 255 itself)."""
    rng = random.Random(3)
    s, rd = make_pile(91, S=3000, coverage=8, e=0.05, min_read=2000, mean_read=2600, sd_read=200)
    pile = [codes_to_str(x) for x in pile_to_seqs(s, rd)]
    seed = pile[0]
    forced = {}
    for k, n in enumerate(runs):
        for rep in range(2):
            at = 700 + 37 * k + 400 * rep + rng.randrange(0, 64)
            block = "".join(rng.choice("ACGT") for _ in range(n))
            pile.append(seed[:at] + block + seed[at:])
            forced[(0, len(pile) - 1)] = (seed[:at] + block + seed[at:], seed[:at] + "-" * n + seed[at:])
    st = D.stage_piles([pile], port, accept_all=True, forced=forced)
    check_stages(st, pile, 2, scores=False)
    G = model_of(st, pile)
    assert max(G.max_delta) == min(max(runs), 254)
    cut = [g for g in forced.values() if len(g[0]) - len(seed) >= 255]
    assert sum(1 for t in range(G.T) if G.cov[t] == 0) == 0 or cut  # (a cut alignment stops covering)
