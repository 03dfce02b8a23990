"""GPU parity tests: the HIP path, called through the C ABI of
include/falcon_amd.h, against (a) the golden vectors generated from the compiled
reference and (b) the CPU oracle on seeded synthetic piles.  Bit-exact."""
import os

import numpy as np
import pytest

from conftest import load_golden
from helpers import check_align_case, check_config_case, check_pile_case, config_pile, sha_ints

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "falcon_amd", "libfalcon_amd.so")

F1 = load_golden("f1_f2_hits_ranges")["cases"]
F3 = load_golden("f3_align")["cases"]
F4 = load_golden("f4_piles")["cases"]


@pytest.fixture(scope="module")
def legacy():
    from oracle.pyoracle import LegacyABI
    return LegacyABI(SO)


@pytest.fixture(scope="module")
def engine():
    from falcon_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_native_library_is_loaded(engine):
    maps = open("/proc/self/maps").read()
    assert "libfalcon_amd.so" in maps


@pytest.mark.parametrize("case", F3, ids=["%s_b%d" % (c["name"], c["band"]) for c in F3])
def test_align_golden_legacy_abi(legacy, case):
    """Every golden alignment, band 150 (k_align.hip) and the contig-layout band 1500
    (k_align_wide.hip), through the legacy `align` symbol."""
    check_align_case(legacy, case)


def test_align_golden_one_launch(engine):
    cases = [c for c in F3 if c["band"] == 150 and c["want_str"]]
    res = engine.align_pairs([(c["q"], c["t"]) for c in cases], band=150, want_str=True)
    for c, r in zip(cases, res):
        for k, v in c["expect"].items():
            assert r[k] == v, (c["name"], k)


def test_absurd_band_is_refused_loudly(engine):
    from falcon_amd.lib import FalconAmdError
    with pytest.raises(FalconAmdError):
        engine.align_pairs([("ACGT" * 10, "ACGT" * 10)], band=5000)


def test_limits_are_errors_not_silent_degradation(engine):
    """K != 8, a seed of >= 100 000 bases (the reference asserts, falcon.c:343), bytes other
    than upper-case ACGT (the reference aligns raw characters and codes other bytes 0xff / 0
    in its k-mer tables, kmer_lookup.c:159-171,236-249: outside the parity domain): loud
    errors naming the offender, never a silently different answer."""
    from falcon_amd.lib import FalconAmdError
    rng = np.random.default_rng(2)
    seq = "".join("ACGT"[i] for i in rng.integers(0, 4, 1200))
    b = engine.batch([[seq, seq, seq]])
    with pytest.raises(FalconAmdError, match="K=9"):
        b.run(4, 9, 0.70)
    b.free()
    big = "".join("ACGT"[i] for i in rng.integers(0, 4, 100000))
    with pytest.raises(FalconAmdError, match="100000"):
        engine.batch([[big, big[:5000]]])
    for bad, at in (("N", 700), ("a", 0), ("\r", 1199)):
        dirty = seq[:at] + bad + seq[at + 1:]
        # the legacy align() has no way to report one pair: the call fails, naming the byte
        with pytest.raises(FalconAmdError, match="holds byte 0x%02x at position %d" % (ord(bad), at)):
            engine.align_pairs([(seq, seq), (dirty, seq)], band=150)
    b = engine.batch([[seq, seq, seq]])  # the context is fine afterwards
    b.run(4, 8, 0.70)
    b.free()


def test_a_dirty_pile_fails_alone(engine, port):
    """A byte other than upper-case ACGT (the reference aligns raw characters there: outside
    the parity domain) does not take its batch down: that pile alone gets no consensus and a
    reason naming sequence, byte and position (fa_batch_pile_error code 3); the piles around
    it are corrected as if it were not there -- whether the byte sits in a read or in the seed."""
    clean = [_synthetic(61, S=5000, coverage=14), _synthetic(62, S=4000, coverage=12)]
    want = [port.generate_consensus(p, 4, 8, 0.70) for p in clean]
    for where, bad, at in ((3, "N", 700), (0, "a", 0), (2, "\r", 1500)):
        dirty = list(_synthetic(63, S=4500, coverage=12))
        dirty[where] = dirty[where][:at] + bad + dirty[where][at + 1:]
        b = engine.batch([clean[0], dirty, clean[1]])
        try:
            b.run(4, 8, 0.70).fetch(True)
            fails = b.failures()
            assert len(fails) == 1 and fails[0][0] == 1
            assert "sequence %d of pile 1 holds byte 0x%02x at position %d" % (where, ord(bad), at) in fails[0][1]
            assert b.stats().n_piles_failed == 1
            assert b.result(1)[0] == ""
            assert tuple(b.result(0)) == tuple(want[0]) and tuple(b.result(2)) == tuple(want[1])
        finally:
            b.free()


def test_wide_band_vs_oracle(engine):
    """band_tolerance 1500 (graph_to_contig.py:52-105): divergent pairs whose band grows
    far past the tuned kernel's 190 diagonals, a pair with a 700-base indel (only a wide
    band crosses it), identical and unalignable pairs -- against the oracle, strings
    included, several pairs in one launch."""
    from oracle.pyoracle import Port
    from falcon_amd.synth import codes_to_str, noisy
    port = Port()
    rng = np.random.default_rng(3)
    pairs = []
    for n, e in [(3000, 0.20), (6000, 0.25), (12000, 0.15), (2500, 0.30)]:
        t = rng.integers(0, 4, n).astype(np.uint8)
        pairs.append((codes_to_str(noisy(t, rng, e)), codes_to_str(t)))
    t = rng.integers(0, 4, 9000).astype(np.uint8)
    q = np.concatenate([t[:4000], rng.integers(0, 4, 700).astype(np.uint8), t[4000:]])
    pairs.append((codes_to_str(noisy(q, rng, 0.05)), codes_to_str(t)))
    pairs.append((pairs[0][1], pairs[0][1]))
    pairs.append((codes_to_str(rng.integers(0, 4, 2000).astype(np.uint8)),
                  codes_to_str(rng.integers(0, 4, 2000).astype(np.uint8))))
    got = engine.align_pairs(pairs, band=1500, want_str=True)
    widest = 0
    for (q, t), r in zip(pairs, got):
        want = port.align(q, t, 1500, 1)
        for k in ("aln_str_size", "dist", "aln_q_s", "aln_q_e", "aln_t_s", "aln_t_e",
                  "q_aln_str", "t_aln_str"):
            assert r[k] == want[k], k
        widest = max(widest, want["dist"])
    assert widest > 1500  # rows far wider than the 190-diagonal kernel could hold


def test_align_beyond_100kb_band_1500(engine, port):
    """graph_to_contig.get_aln_data aligns overlapping contig ends with band 1500 over up to
    250 kb (falcon_kit/mains/graph_to_contig.py:52-105): 120 kb at 2 % and 230 kb at 0.3 %
    divergence through k_align_wide against the oracle -- the six numbers, and the gapped
    strings by digest."""
    import hashlib
    import time
    from falcon_amd.synth import codes_to_str, noisy
    rng = np.random.default_rng(77)
    pairs = []
    for n, e in ((120000, 0.02), (230000, 0.003)):
        t = rng.integers(0, 4, n).astype(np.uint8)
        pairs.append((codes_to_str(noisy(t, rng, e)), codes_to_str(t)))
    t0 = time.time()
    got = engine.align_pairs(pairs, band=1500, want_str=True)
    dt = time.time() - t0
    for (q, t), r in zip(pairs, got):
        want = port.align(q, t, 1500, 1)
        for k in ("aln_str_size", "dist", "aln_q_s", "aln_q_e", "aln_t_s", "aln_t_e"):
            assert r[k] == want[k], k
        assert r["aln_str_size"] > 100000
        for k in ("q_aln_str", "t_aln_str"):
            assert hashlib.sha1(r[k].encode()).hexdigest() == hashlib.sha1(want[k].encode()).hexdigest(), k
    print("align_pairs of 120 kb + 230 kb at band 1500: %.2f s (staging, kernel, strings)" % dt)


def test_chain_ranges_golden(engine):
    """k_seed_index + k_chain vs find_kmer_pos_for_seq + find_best_aln_range."""
    cases = [c for c in F1 if c["mask"] < 0]
    b = engine.batch([[c["seed"], c["query"]] for c in cases])
    b.run(4, 8, 0.70)
    for i, c in enumerate(cases):
        r = b.range(2 * i + 1)
        assert r["n_hit"] == c["count"], c["name"]
        assert [r["s1"], r["e1"], r["s2"], r["e2"], r["score"]] == c["range_48_5"], c["name"]
    b.free()


def _planted(rng, runs, seed_len, query_len, homopolymer_at=()):
    """Seed and query in which the only 8-mer matches (up to accidents, which both sides see
    alike) are planted: `runs` = [(q0, t0, n_hits)]: n_hits consecutive probes (every 4th
    query offset) on one diagonal.  Fillers come from disjoint alphabets."""
    seed = [rng.choice("AC") for _ in range(seed_len)]
    query = [rng.choice("GT") for _ in range(query_len)]
    for q0, t0, n in runs:
        word = [rng.choice("ACGT") for _ in range(4 * (n - 1) + 8)]
        seed[t0:t0 + len(word)] = word
        query[q0:q0 + len(word)] = word
    for q0, t0, reps in homopolymer_at:   # one query 8-mer, `reps` seed positions in a row
        seed[t0:t0 + 8 + reps - 1] = "G" * (8 + reps - 1)
        query[q0:q0 + 8] = "G" * 8
    return "".join(seed)[:seed_len], "".join(query)[:query_len]


def test_chain_tie_breaks_on_planted_hits(engine, port):
    """The tie-breaks of find_best_aln_range (kmer_lookup.c:360-366: first fullest bin in hit
    order; :396-410: the running score resets at gaps, repeated query positions count +32
    each) on k_chain itself: planted hit patterns in the style of tests/golden/f2_ranges_extra
    (equal counts in two bins in both orders, gap resets, a query 8-mer hitting a run of
    seed positions, counts around the threshold of 5), k_seed_index + k_chain against the
    oracle's hit list and window on the same strings."""
    import random
    rng = random.Random(2024)
    designs = []
    for _ in range(12):   # two runs of equal length on different diagonals, either order
        n = rng.choice([6, 7, 9, 12])
        qa, qb = sorted(rng.sample(range(0, 1500, 4), 2))
        if qb - qa < 4 * n + 40:
            qb = qa + 4 * n + 40 + 4 * rng.randint(0, 50)
        ta, tb = rng.sample([200, 900, 1700, 2600, 3300], 2)
        designs.append(([(qa, ta, n), (qb, tb, n)], ()))
    for _ in range(8):    # one diagonal broken by gaps: the score resets (:396-410)
        n1, n2, n3 = rng.choice([3, 6, 8]), rng.choice([6, 7]), rng.choice([2, 6, 10])
        q0, gap1, gap2 = 4 * rng.randint(0, 30), 4 * rng.randint(20, 120), 4 * rng.randint(20, 120)
        q1 = q0 + 4 * n1 + gap1
        q2 = q1 + 4 * n2 + gap2
        designs.append(([(q0, q0 + 300, n1), (q1, q1 + 300, n2), (q2, q2 + 300, n3)], ()))
    for _ in range(8):    # counts around the threshold (count_th = 5: more than 5 hits needed)
        n = rng.choice([4, 5, 6])
        designs.append(([(4 * rng.randint(0, 100), rng.randint(0, 2000), n)], ()))
    for _ in range(8):    # a query 8-mer on 2..6 consecutive seed positions inside a run
        n = rng.choice([6, 8])
        q0, t0 = 4 * rng.randint(5, 60), rng.randint(100, 1500)
        designs.append(([(q0, t0, n), (q0 + 4 * n + 8, t0 + 4 * n + 8 + rng.choice([0, 3]), n)],
                        [(q0 + 4 * n, t0 + 4 * n - rng.randint(0, 2), rng.randint(2, 6))]))
    pairs = [_planted(rng, runs, 4000, 2400, homo) for runs, homo in designs]
    b = engine.batch([[seed, query] for seed, query in pairs])
    try:
        b.run(4, 8, 0.70)
        interesting = 0
        for i, (seed, query) in enumerate(pairs):
            hq, ht = port.find_hits(seed, query)
            want = list(port.best_range(hq, ht, 48, 5))
            r = b.range(2 * i + 1)
            assert r["n_hit"] == len(hq), i
            assert [r["s1"], r["e1"], r["s2"], r["e2"], r["score"]] == want, (i, designs[i])
            interesting += want != [0, 0, 0, 0, 0]
        assert interesting >= 25   # most designs produce a window, the sub-threshold ones none
    finally:
        b.free()


@pytest.mark.parametrize("case", F4, ids=[c["name"] for c in F4])
def test_piles_golden_legacy_abi(legacy, case):
    check_pile_case(legacy, case)


def test_piles_golden_one_batch(engine):
    by_cfg = {}
    for c in F4:
        by_cfg.setdefault((c["min_cov"], c["min_idt"]), []).append(c)
    for (min_cov, min_idt), cases in by_cfg.items():
        res = engine.consensus([c["seqs"] for c in cases], min_cov, 8, min_idt, want_eqv=True)
        for c, (seq, eqv) in zip(cases, res):
            assert seq == c["sequence"], c["name"]
            assert sha_ints(eqv) == c["eqv_sha"], c["name"]


F8 = load_golden("f8_configs")["cases"]


@pytest.mark.parametrize("case", F8, ids=[c["name"] for c in F8])
def test_full_size_config_piles_golden(legacy, case):
    """BASELINE configs 2, 4 (dmel-like, 200 reads of a 30 kb seed at 80x) and 5
    (Arabidopsis-like, two haplotypes) at full size against the compiled reference's
    answer, through the legacy C ABI."""
    check_config_case(legacy, case)


def test_full_size_config_piles_one_batch(engine):
    """... and together in one batch (mixed seed lengths and depths side by side)."""
    piles = [config_pile(c) for c in F8]
    got = engine.consensus(piles + piles[::-1], 4, 8, 0.70, want_eqv=True)
    for c, (seq, eqv) in zip(F8 + F8[::-1], got):
        assert seq == c["sequence"], c["name"]
        assert sha_ints(eqv) == c["eqv_sha"], c["name"]


def _synthetic(seed, max_n_read=200, **kw):
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    s, rd = make_pile(seed, **kw)
    return [codes_to_str(x) for x in pile_to_seqs(s, rd, max_n_read)]


def test_synthetic_piles_vs_oracle(engine, port):
    piles = [
        _synthetic(31, S=3000, coverage=12, min_read=500, mean_read=1500, sd_read=500),
        _synthetic(32, S=8000, coverage=25, min_read=1000, mean_read=5000, sd_read=2000),
        _synthetic(33, S=8000, coverage=30, het=0.005, min_read=1000, mean_read=5000, sd_read=2000),
        _synthetic(34, S=6000, coverage=20, e=0.20, min_read=1000, mean_read=4000, sd_read=1500),
        _synthetic(35, S=6000, coverage=20, e=0.02, min_read=1000, mean_read=4000, sd_read=1500),
    ]
    for min_cov, idt in ((4, 0.70), (0, 0.80), (8, 0.70)):
        got = engine.consensus(piles, min_cov, 8, idt, want_eqv=True)
        for p, (seq, eqv) in zip(piles, got):
            eseq, eeqv = port.generate_consensus(p, min_cov, 8, idt)
            assert seq == eseq
            assert eqv == eeqv


def test_stage_outputs_vs_oracle(engine, port):
    """Per-read stage outputs (range, alignment summary, cell counts) and the
    B_alg work statistics agree with the oracle."""
    pile = _synthetic(41, S=5000, coverage=15, min_read=800, mean_read=3000, sd_read=1000)
    b = engine.batch([pile])
    b.run(4, 8, 0.70)
    _seq, _eqv, st = port.generate_consensus(pile, 4, 8, 0.70, want_stats=True)
    gs = b.stats()
    assert (gs.L, gs.C, gs.D, gs.A, gs.T, gs.O, gs.n_aligned) == \
        (st["L"], st["C"], st["D"], st["A"], st["T"], st["O"], st["n_aligned"])
    for g in range(1, len(pile)):
        hq, ht = port.find_hits(pile[0], pile[g])
        r = b.range(g)
        assert [r["s1"], r["e1"], r["s2"], r["e2"], r["score"]] == list(port.best_range(hq, ht))
        if r["ok"]:
            a = port.align(pile[g][r["s1"]:r["e1"]], pile[0][r["s2"]:r["e2"]], 150, 1)
            ga = b.alignment(g)
            assert (ga["dist"], ga["q_e"], ga["t_e"], ga["size"], ga["cells"]) == \
                (a["dist"], a["aln_q_e"], a["aln_t_e"], a["aln_str_size"], a["cells"])
    b.free()


def _reference_tags(q_aln, t_aln):
    """The tag list of one alignment as get_align_tags builds it (src/c/falcon.c:106-162), seed
    positions counted from the alignment's first one: every column yields (t, delta, base) --
    t the last seed position consumed (-1 before the first), delta the number of read bases
    inserted behind it so far, base the read's character ('-' where it skips the seed base);
    tagging stops where an insertion run reaches 255."""
    out, t, delta, prev_delta = [], -1, 0, 0
    for qc, tc in zip(q_aln, t_aln):
        if qc != "-":
            delta += 1
        if tc != "-":
            t += 1
            delta = 0
        if delta >= 255 or prev_delta >= 255:
            break
        out.append((t, delta, qc))
        prev_delta = delta
    return out


def test_hit_lists_and_tag_words_vs_reference(engine, port):
    """Two intermediate lists compared directly (fa_batch_debug_hits / _tags), not through what
    later stages make of them: every read's k-mer hit list in the reference's order
    (kmer_lookup.c:207-286) as k_chain enumerates it, and every accepted alignment's tags
    (falcon.c:106-162) as k_tags packs them into its per-position words."""
    piles = [_synthetic(43, S=5000, coverage=15, min_read=800, mean_read=3000, sd_read=1000),
             _synthetic(44, S=9000, coverage=25),
             _synthetic(45, S=4000, coverage=12, min_read=2500, mean_read=3500, sd_read=300)]
    # (insertion runs of more than 11 bases leave the tag word for the alignment's byte list)
    rng = np.random.default_rng(45)
    for j, n in ((1, 20), (2, 40), (3, 13)):
        r = piles[2][j]
        piles[2][j] = r[:len(r) // 2] + "".join("ACGT"[c] for c in rng.integers(0, 4, n)) + r[len(r) // 2:]
    b = engine.batch(piles)
    b.run(4, 8, 0.70)
    g0, n_hits, n_tags, n_long = 0, 0, 0, 0
    for pile in piles:
        seed = pile[0]
        assert b.debug_hits(g0) == ([], [])  # (the seed is the target, not a query)
        for j in range(1, len(pile)):
            g = g0 + j
            hq, ht = port.find_hits(seed, pile[j])
            gq, gt = b.debug_hits(g)
            assert (gq, gt) == (hq, ht), (g, len(gq), len(hq))
            n_hits += len(hq)
            r, ga = b.range(g), b.alignment(g)
            tags = b.debug_tags(g)
            if not (r["ok"] and ga["accept"]):
                assert tags == []
                continue
            a = port.align(pile[j][r["s1"]:r["e1"]], seed[r["s2"]:r["e2"]], 150, 1)
            want = _reference_tags(a["q_aln_str"], a["t_aln_str"])
            got = [(t, d, seed[r["s2"] + t] if c is None else c) for t, d, c in tags]
            assert got == want, (g, len(got), len(want))
            n_tags += len(want)
            n_long += sum(1 for t, d, c in want if d > 11)
        g0 += len(pile)
    b.free()
    assert n_hits > 20000 and n_tags > 250000 and n_long > 0


def test_seed_index_in_the_lds_and_the_kernel_behind_it(engine, port, monkeypatch):
    """k_seed_index keeps a pile's 8-mer table in the LDS with 16-bit cursors, which holds seeds of up to
    65 543 bases; longer ones (< 100 000) go to k_seed_index_long in the same launch sequence.  One batch
    with seeds of 6 000, exactly 65 543, 65 544 and 80 000 bases plus a low-complexity seed (8-mers many
    times inside 64 consecutive positions): every read's hit list in the reference's order
    (kmer_lookup.c:207-286) -- it is read straight off the index -- and the consensus; then the golden
    piles with every pile forced through the kernel behind (FALCON_AMD_INDEX_LONG)."""
    rng = np.random.default_rng(61)

    def pile_on(seed_len, n_reads, read_len, lowc=False):
        if lowc:
            unit = "".join("ACGT"[c] for c in rng.integers(0, 4, 5))
            seed = "".join(("ACGT"[c] if rng.random() < 0.03 else unit[i % 5]) for i, c in
                           enumerate(rng.integers(0, 4, seed_len)))
        else:
            seed = "".join("ACGT"[c] for c in rng.integers(0, 4, seed_len))
        reads = []
        for _ in range(n_reads):
            at = int(rng.integers(0, max(1, seed_len - read_len)))
            r = [c for c in seed[at:at + read_len] if rng.random() > 0.04]
            reads.append("".join(c if rng.random() > 0.05 else "ACGT"[rng.integers(0, 4)] for c in r))
        return [seed, seed] + reads
    piles = [pile_on(6000, 8, 3000), pile_on(65543, 6, 5000), pile_on(65544, 6, 5000), pile_on(80000, 6, 5000),
             pile_on(3000, 8, 1500, lowc=True)]
    b = engine.batch(piles)
    b.run(2, 8, 0.70)
    g0, n_hits = 0, 0
    for p, pile in enumerate(piles):
        for j in range(1, len(pile)):
            hq, ht = port.find_hits(pile[0], pile[j])
            assert b.debug_hits(g0 + j) == (hq, ht), (p, j)
            n_hits += len(hq)
        g0 += len(pile)
    b.free()
    assert n_hits > 30000
    for p, got in zip(piles, engine.consensus(piles, 2, 8, 0.70, want_eqv=True)):
        assert got == tuple(port.generate_consensus(p, 2, 8, 0.70))

    class Impl:
        def generate_consensus(self, seqs, min_cov, K, min_idt):
            return engine.consensus([seqs], min_cov, K, min_idt, want_eqv=True)[0]
    monkeypatch.setenv("FALCON_AMD_INDEX_LONG", "1")
    for c in F4:
        check_pile_case(Impl(), c)


def test_device_side_packing_route_gives_the_same_answers(engine):
    """Batches are packed to 2 bits per base on the host (pack_host.cpp); round 2's route --
    the text uploaded and packed by k_pack -- is still there behind FALCON_AMD_DEVICE_PACK (the
    counter calibration of the traffic measurement runs on k_pack): same consensus either way,
    a dirty pile failing alone either way."""
    import json
    import subprocess
    import sys
    piles = [_synthetic(46, S=4000, coverage=12, min_read=800, mean_read=2500, sd_read=800),
             _synthetic(47, S=6000, coverage=20)]
    dirty = list(piles[0])
    dirty[3] = dirty[3][:100] + "N" + dirty[3][101:]
    here = engine.consensus(piles + [dirty], 4, 8, 0.70)
    code = ("import json,sys; from falcon_amd.engine import Engine; piles=json.load(sys.stdin); e=Engine(0);"
            "print(json.dumps([str(x) for x in e.consensus(piles,4,8,0.70)])); e.close()")
    out = subprocess.run([sys.executable, "-c", code], input=json.dumps(piles + [dirty]), capture_output=True,
                         text=True, cwd=os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."),
                         env=dict(os.environ, FALCON_AMD_DEVICE_PACK="1"), check=True)
    there = json.loads(out.stdout.strip().splitlines()[-1])
    assert [str(x) for x in here] == there
    from falcon_amd.engine import FailedPile
    assert len(here[0]) > 3000 and len(here[1]) > 5000 and isinstance(here[2], FailedPile) and "0x4e" in here[2].reason


def test_ecoli_scale_pile_vs_oracle(engine, port):
    """BASELINE config 2 shape: ~20 kb seed x 40x."""
    pile = _synthetic(7, S=20000, coverage=40)
    (seq, eqv), = engine.consensus([pile], 4, 8, 0.70, want_eqv=True)
    eseq, eeqv = port.generate_consensus(pile, 4, 8, 0.70)
    assert len(seq) > 19000
    assert seq == eseq and eqv == eeqv


def test_many_reads_pile_uses_wide_chunks(engine, port):
    """> 64 and > 128 accepted alignments exercise the 2- and 4-chunk sweeps."""
    for seed, cov in ((51, 90), (52, 170)):
        pile = _synthetic(seed, S=3000, coverage=cov, min_read=1500, mean_read=2500,
                          sd_read=300, max_n_read=500)
        (seq, eqv), = engine.consensus([pile], 4, 8, 0.70, want_eqv=True)
        eseq, eeqv, st = port.generate_consensus(pile, 4, 8, 0.70, want_stats=True)
        assert st["n_aligned"] > (64 if cov == 90 else 128)
        assert seq == eseq and eqv == eeqv


def test_batch_is_order_preserving_and_rerunnable(engine):
    piles = [_synthetic(60 + i, S=2500, coverage=12, min_read=500, mean_read=1500, sd_read=400)
             for i in range(6)]
    a = engine.consensus(piles, 4, 8, 0.70)
    b = engine.consensus(list(reversed(piles)), 4, 8, 0.70)
    assert a == list(reversed(b))
    bt = engine.batch(piles)
    r1 = [bt.run(4, 8, 0.70).fetch().result(i) for i in range(len(piles))]
    r2 = [bt.run(4, 8, 0.70).fetch().result(i) for i in range(len(piles))]
    assert r1 == r2 == a
    bt.free()


def test_a_batch_beyond_32_bit_base_indices_is_cut_into_launches_of_k_align2(engine, monkeypatch):
    """k_align2 indexes bases with 32 bits: a batch of 2^28 packed words (4.29 G bases) or more is aligned in
    several launches, each on a stretch of whole piles, not handed to the slower k_align (rounds 1-5).
    FALCON_AMD_A2_MAX_WORDS forces the cut on a small batch: same strings, same eqv, the two-per-wavefront
    kernel's iteration counters show it ran, and nothing was handed back."""
    piles = [_synthetic(160 + i, S=2500 + 300 * (i % 3), coverage=12, min_read=500, mean_read=1500, sd_read=400)
             for i in range(9)]
    want = engine.consensus(piles, 4, 8, 0.70, want_eqv=True)
    for lim in ("3000", "9000", "64"):   # (64: every pile a launch of its own)
        monkeypatch.setenv("FALCON_AMD_A2_MAX_WORDS", lim)
        bt = engine.batch(piles)
        bt.run(4, 8, 0.70).fetch(True)
        got = [bt.result(i) for i in range(len(piles))]
        st = bt.stats()
        bt.free()
        assert got == want, lim
        assert st.align_pair_iterations + st.align_single_iterations > 0 and st.align_handed_back == 0, lim
    monkeypatch.delenv("FALCON_AMD_A2_MAX_WORDS")


def test_deep_piles_and_failures_stay_with_their_pile(engine, port):
    """The reference loops over any n_seq (falcon.c:597-647); its driver's default
    --max-n-read is 500.  Piles of ~700 and ~1000 usable reads (more than 64, 128, 256 and 512
    alignments over a segment) and of ~1200 (past the 1023 that rounds 1-3 stopped at: k_links2
    walks any number of alignments, link counts are 16 bits) equal the oracle, in one batch
    with ordinary piles, through the default kernels and through the ones behind them."""
    from falcon_amd.engine import FailedPile
    normal = [_synthetic(700 + i, S=3000, coverage=14, min_read=600, mean_read=1800, sd_read=500)
              for i in range(2)]
    deep = _synthetic(710, max_n_read=5000, S=3000, coverage=440, e=0.10, min_read=1500, mean_read=2400, sd_read=300)
    deeper = _synthetic(711, max_n_read=5000, S=2500, coverage=690, e=0.08, min_read=1500, mean_read=2200, sd_read=200)
    too_deep = _synthetic(712, max_n_read=5000, S=2500, coverage=830, e=0.08, min_read=1500, mean_read=2200,
                          sd_read=200)
    assert 650 < len(deep) < 800 and 900 < len(deeper) <= 1020 and len(too_deep) > 1150
    piles = [normal[0], deep, too_deep, deeper, normal[1]]
    b = engine.batch(piles)
    try:
        b.run(4, 8, 0.70).fetch(True)
        got = [b.result(i) for i in range(len(piles))]
        st = b.stats()
    finally:
        b.free()
    assert st.n_piles_failed == 0
    want = [port.generate_consensus(p, 4, 8, 0.70) for p in piles]
    for i in range(len(piles)):
        assert got[i][0] == want[i][0] and got[i][1] == want[i][1], i
        assert not isinstance(got[i][0], FailedPile)
    # k_links (lanes = alignments, up to 1024 over a segment) instead of k_links2: the piles it holds
    import os
    os.environ["FALCON_AMD_LINKS1"] = "1"
    try:
        for i, r in zip((1, 3), engine.consensus([deep, deeper], 4, 8, 0.70, want_eqv=True)):
            assert r[0] == want[i][0] and r[1] == want[i][1], i
    finally:
        del os.environ["FALCON_AMD_LINKS1"]


def test_score_generic_path_alone(engine, monkeypatch):
    """k_score1 (the general score kernel: what k_score2 hands on) on EVERY level of the golden
    piles: same strings and eqv as the default path and the reference (FALCON_AMD_SCORE_GENERIC
    pins it)."""
    class Impl:
        def generate_consensus(self, seqs, min_cov, K, min_idt):
            return engine.consensus([seqs], min_cov, K, min_idt, want_eqv=True)[0]
    fast = [Impl().generate_consensus(c["seqs"], c["min_cov"], c["K"], c["min_idt"]) for c in F4]
    monkeypatch.setenv("FALCON_AMD_SCORE_GENERIC", "1")
    for c, f in zip(F4, fast):
        check_pile_case(Impl(), c)
        assert Impl().generate_consensus(c["seqs"], c["min_cov"], c["K"], c["min_idt"]) == f


def test_fallback_kernels_alone(engine, monkeypatch):
    """The kernels behind the default ones on EVERY pile and segment of the golden piles:
    k_links (lanes = alignments) instead of k_links2 (FALCON_AMD_LINKS1), k_score1 instead of
    k_score2 (FALCON_AMD_SCORE1), and both -- same strings and eqv as the default path and the
    reference.  They take what the default kernels hand on (positions with long insertion runs,
    the unitig mode, piles whose score bound outgrows 32 bits)."""
    class Impl:
        def generate_consensus(self, seqs, min_cov, K, min_idt):
            return engine.consensus([seqs], min_cov, K, min_idt, want_eqv=True)[0]
    fast = [Impl().generate_consensus(c["seqs"], c["min_cov"], c["K"], c["min_idt"]) for c in F4]
    for env in (("FALCON_AMD_LINKS1",), ("FALCON_AMD_SCORE1",), ("FALCON_AMD_LINKS1", "FALCON_AMD_SCORE1")):
        for e in env:
            monkeypatch.setenv(e, "1")
        for c, f in zip(F4, fast):
            check_pile_case(Impl(), c)
            assert Impl().generate_consensus(c["seqs"], c["min_cov"], c["K"], c["min_idt"]) == f
        for e in env:
            monkeypatch.delenv(e)


def test_pipelined_submit_wait_equals_run(engine):
    """fa_batch_submit / fa_batch_wait: two (and three) batches of one context in flight, the
    next one's throughput stages beside the previous one's score recurrence and back-trace
    on their own stream -- same strings, same eqv, same statistics as fa_batch_run; also
    with the batches driven from different threads, and freed while in flight."""
    import threading
    sets = [[_synthetic(300 + 10 * k + i, S=3000 + 500 * k, coverage=14, min_read=600, mean_read=1800,
                        sd_read=500) for i in range(8)] for k in range(3)]
    want, stats = [], []
    for ps in sets:
        b = engine.batch(ps)
        b.run(4, 8, 0.70).fetch(True)
        want.append([b.result(i) for i in range(len(ps))])
        stats.append((b.stats().C, b.stats().D, b.stats().A, b.stats().O))
        b.free()
    bs = [engine.batch(ps) for ps in sets]
    for rounds in range(3):
        bs[0].submit(4, 8, 0.70)
        bs[1].submit(4, 8, 0.70)
        bs[0].wait()
        bs[2].submit(4, 8, 0.70)
        got0 = [bs[0].fetch(True).result(i) for i in range(len(sets[0]))]
        bs[0].submit(4, 8, 0.70)          # again while the other two are still out
        bs[1].wait(); bs[2].wait(); bs[0].wait()
        got = [got0] + [[b.fetch(True).result(i) for i in range(b.n_pile)] for b in bs[1:]]
        assert got == want
        assert [(b.stats().C, b.stats().D, b.stats().A, b.stats().O) for b in bs] == stats
    # one thread per batch on the same context
    out, errs = [None] * 3, []

    def work(k):
        try:
            for _ in range(4):
                bs[k].submit(4, 8, 0.70)
                bs[k].wait()
                out[k] = [bs[k].fetch(True).result(i) for i in range(bs[k].n_pile)]
        except Exception as e:
            errs.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs and out == want
    from falcon_amd.lib import FalconAmdError
    bs[1].submit(4, 8, 0.70)
    with pytest.raises(FalconAmdError, match="still running"):
        bs[1].submit(4, 8, 0.70)
    with pytest.raises(FalconAmdError):
        bs[1].fetch()
    for b in bs:
        b.free()              # (bs[1] is in flight: free waits for its kernels)
    with pytest.raises(FalconAmdError):
        engine.batch(sets[0]).wait()


def test_bench_scale_batch_properties(engine):
    """The bench workload (BASELINE config 2: ~20 kb seeds x 40x, e = 0.13), a batch large
    enough to fill every wave slot several times over: results must not depend on what
    else is in the batch, on the order of the piles, or on the run -- the properties that
    hold at any size (the oracle itself checks one such pile above)."""
    import hashlib
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    piles = []
    for i in range(72):
        s, rd = make_pile(7000 + i, S=20000, coverage=40.0)
        piles.append([codes_to_str(x) for x in pile_to_seqs(s, rd, 200)])
    bt = engine.batch(piles)
    try:
        r1 = [bt.run(4, 8, 0.70).fetch(True).result(i) for i in range(len(piles))]
        st = bt.stats()
        r2 = [bt.run(4, 8, 0.70).fetch(True).result(i) for i in range(len(piles))]
    finally:
        bt.free()
    assert r1 == r2                                     # rerunnable, deterministic
    assert st.n_piles == len(piles) and st.n_aligned > 60 * len(piles)
    # the tape arena (one-byte cells, 80 bytes per iteration) holds this workload: nothing is
    # handed to the general kernel, and a wavefront's slot stays under 3 MB
    assert st.align_relaunched == 0 and st.align_handed_back == 0
    assert st.align_slot_cells // 64 * 80 < 3 * 1024 * 1024
    assert st.align_pair_iterations > st.align_single_iterations // 4  # two tracks do run side by side
    assert all(len(c) > 19000 for c, _ in r1)           # every seed corrected end to end
    rev = engine.consensus(list(reversed(piles)), 4, 8, 0.70, want_eqv=True)
    assert rev == list(reversed(r1))                    # order of piles is irrelevant
    for i in (0, 31, 71):                               # and so is their company
        assert engine.consensus([piles[i]], 4, 8, 0.70, want_eqv=True) == [r1[i]]
    # ... and every one of the 72 is the COMPILED REFERENCE's answer (tests/golden/f10_bench72,
    # oracle/gen_golden.py f10): input, consensus string and eqv digests per pile
    from helpers import sha_ints
    f10 = load_golden("f10_bench72")["cases"]
    assert [c["seed"] for c in f10] == list(range(7000, 7072))
    for i, (c, (cns, eqv)) in enumerate(zip(f10, r1)):
        assert hashlib.sha1("\n".join(piles[i]).encode()).hexdigest()[:16] == c["input_sha"], i
        assert (len(cns), hashlib.sha1(cns.encode()).hexdigest(), sha_ints(eqv)) == \
               (c["cns_len"], c["cns_sha"], c["eqv_sha"]), i


@pytest.mark.parametrize("kernel", ["two_per_wave", "one_per_wave"])
def test_outgrown_alignment_slots_are_redone(monkeypatch, port, kernel):
    """The alignment arena is sized for what alignments use, not for the worst case: an
    alignment that outgrows what it was given is reported by the kernel (nothing is written
    past a slot) and done again, alone, by the general kernel in one of a few worst-case
    slots.  Forced here: k_align2 with a tape ring of 1024 iterations (alignments of more
    than ~1700 rows are refused at the queue or handed back when their tape is used up),
    k_align (FALCON_AMD_ALIGN1) with slots of 2 diagonals per row.  Same results as the
    oracle, batch after batch."""
    from falcon_amd.engine import Engine
    if kernel == "two_per_wave":
        monkeypatch.setenv("FALCON_AMD_RING", "1024")
        piles = [
            _synthetic(32, S=12000, coverage=25, min_read=1000, mean_read=7000, sd_read=3000),
            _synthetic(34, S=9000, coverage=20, e=0.20, min_read=1000, mean_read=6000, sd_read=2500),
        ]
    else:
        monkeypatch.setenv("FALCON_AMD_ALIGN1", "1")
        monkeypatch.setenv("FALCON_AMD_SLOT_WIDTH", "2")
        piles = [
            _synthetic(32, S=8000, coverage=25, min_read=1000, mean_read=5000, sd_read=2000),
            _synthetic(34, S=6000, coverage=20, e=0.20, min_read=1000, mean_read=4000, sd_read=1500),
        ]
    want = [port.generate_consensus(p, 4, 8, 0.70) for p in piles]
    eng = Engine(0)
    try:
        b = eng.batch(piles)
        b.run(4, 8, 0.70).fetch(True)
        st = b.stats()
        got = [b.result(i) for i in range(len(piles))]
        b.free()
        assert 10 < st.align_relaunched <= st.n_seqs
        assert [tuple(x) for x in got] == [tuple(x) for x in want]
        b = eng.batch(piles[::-1])
        b.run(4, 8, 0.70).fetch(True)
        st2 = b.stats()
        got2 = [b.result(i) for i in range(len(piles))]
        b.free()
        if kernel == "two_per_wave":
            # (which alignments share a wavefront, and so how long one waits on the tape beside its
            # neighbour, depends on the order the wavefronts take them off the queue: the count is
            # not a function of the input alone)
            assert abs(st2.align_relaunched - st.align_relaunched) <= max(3, st.align_relaunched // 8)
        else:
            assert st2.align_relaunched == st.align_relaunched and st2.align_slot_cells == st.align_slot_cells
        assert [tuple(x) for x in got2] == [tuple(x) for x in want[::-1]]
    finally:
        eng.close()


@pytest.mark.parametrize("cause", ["tape", "wide_rows", "escape_list"])
def test_every_hand_back_cause_on_the_device(monkeypatch, port, cause):
    """k_align2's three ways of giving an alignment back to the general kernel, each FORCED on
    the GPU and counted by cause (fa_stats.align_handed_back_*) -- the lane emulator proves their
    control logic, only the device runs the hand-scheduled row tail and the packed state calls
    around them:
      tape         a tape ring of 1024 iterations under alignments of up to ~3000 rows;
      wide_rows    no patience with a neighbour whose band is wider than 60 diagonals (two piles
                   of the bench workload: a fifth of their alignments meets such rows);
      escape_list  reads that copy the seed but for a base every ~300: snakes of >= 255 bases in
                   every row, and a list that is declared full at 16 entries.
    The consensus of every pile equals the oracle's (DW_banded.c:183-243)."""
    from falcon_amd.engine import Engine
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    import random
    if cause == "tape":
        monkeypatch.setenv("FALCON_AMD_RING", "1024")
        piles = [_synthetic(41, S=12000, coverage=20, min_read=1000, mean_read=7000, sd_read=3000)]
    elif cause == "wide_rows":
        # (a fifth of the bench workload's alignments meets rows of more than 60 diagonals for a
        # few rows: with no patience at all beside a waiting neighbour, those go back)
        monkeypatch.setenv("FALCON_AMD_WIDE_PATIENCE", "1")
        piles = [_synthetic(42 + i, S=20000, coverage=40) for i in range(2)]
    else:
        monkeypatch.setenv("FALCON_AMD_ESC_CAP", "16")
        rng = random.Random(9)
        s, _rd = make_pile(44, S=30000, coverage=1, e=0.0)
        seed = codes_to_str(s)
        reads = []
        for k in range(24):
            r = list(seed[rng.randrange(0, 3000):rng.randrange(24000, 30000)])
            for at in range(rng.randrange(150, 300), len(r) - 10, rng.randrange(280, 330)):
                r[at] = "ACGT"[("ACGT".index(r[at]) + 1 + rng.randrange(3)) % 4]
            reads.append("".join(r))
        piles = [[seed, seed] + reads]
    want = [port.generate_consensus(p, 4, 8, 0.70) for p in piles]
    eng = Engine(0)
    try:
        b = eng.batch(piles)
        b.run(4, 8, 0.70).fetch(True)
        st = b.stats()
        got = [b.result(i) for i in range(len(piles))]
        b.free()
    finally:
        eng.close()
    by = {"tape": st.align_handed_back_tape, "wide_rows": st.align_handed_back_wide,
          "escape_list": st.align_handed_back_escapes}
    assert by[cause] > 0, (by, st.align_handed_back)
    assert st.align_handed_back == sum(by.values()) == st.align_relaunched
    assert [tuple(x) for x in got] == [tuple(x) for x in want]


def test_long_insertion_runs_and_tag_cutoff(engine, port):
    """Insertion runs longer than a tag's 16 inline bases, and runs past the
    reference's 255-column cut-off (falcon.c:138-152; beyond the reference's own
    parity domain, so the oracle is the arbiter there)."""
    from test_oracle_vs_ref import _pile_with_long_insertions
    for runs in ([20, 40, 120, 200], [17, 300, 260]):
        pile = _pile_with_long_insertions(runs)
        (seq, eqv), = engine.consensus([pile], 2, 8, 0.70, want_eqv=True)
        eseq, eeqv = port.generate_consensus(pile, 2, 8, 0.70)
        assert len(seq) > 2500
        assert seq == eseq and eqv == eeqv


# --------------------------------------------------------------------------
# --trim windows on the GPU: mask_k_mer + find_kmer_pos_for_seq +
# find_best_aln_range2 of consensus.py:48-99, batched (k_trimwin.hip)
# --------------------------------------------------------------------------
def _trim_ranges(engine, piles, mask=16):
    b = engine.batch(piles)
    try:
        b.trim_windows(8, mask)
        out = []
        g = 0
        for pile in piles:
            rows = []
            for j in range(len(pile)):
                r = b.range(g)
                rows.append(([r["s1"], r["e1"], r["s2"], r["e2"], r["score"]], r["n_hit"]))
                g += 1
            out.append(rows)
        return out
    finally:
        b.free()


def test_trim_windows_golden(engine):
    """Every golden case that carries the reference's own find_best_aln_range2 answer
    (masked hits, mask = 16), all in one batch: pile = [seed, query]."""
    cases = [c for c in F1 if "range2" in c and c["mask"] == 16]
    assert cases
    got = _trim_ranges(engine, [[c["seed"], c["query"]] for c in cases])
    for c, rows in zip(cases, got):
        assert rows[1][0] == c["range2"], c["name"]
        assert rows[1][1] == c["count"], c["name"]


def test_trim_windows_vs_oracle(engine):
    """Seeded synthetic piles (noisy reads, partial overlaps, a repeat-rich seed whose
    frequent k-mers get masked, reads with more hits than fit LDS) against the oracle."""
    from oracle.pyoracle import Port
    from falcon_amd.synth import codes_to_str, make_pile
    port = Port()
    rng = np.random.default_rng(11)
    piles = []
    for s, (S, cov) in enumerate([(3000, 12.0), (9000, 10.0), (20000, 6.0)]):
        seed, reads = make_pile(500 + s, S=S, coverage=cov, mean_read=min(S, 6000) * 0.7,
                                sd_read=600, min_read=1200)
        piles.append([codes_to_str(seed)] + [codes_to_str(x) for x in reads])
    # repeat-rich seed: a 40-mer tiled with mutations -> many k-mers above the mask threshold
    unit = "".join("ACGT"[i] for i in rng.integers(0, 4, 40))
    rep = []
    for _ in range(120):
        u = list(unit)
        for _ in range(2):
            u[int(rng.integers(0, 40))] = "ACGT"[int(rng.integers(0, 4))]
        rep.append("".join(u))
    rep_seed = "".join(rep)
    rep_reads = [rep_seed[a:a + 2500] for a in (0, 700, 1500, 2200)]
    piles.append([rep_seed] + rep_reads + ["".join("ACGT"[i] for i in rng.integers(0, 4, 1800))])
    got = _trim_ranges(engine, piles)
    n_window = 0
    for pile, rows in zip(piles, got):
        for j in range(1, len(pile)):
            q, t = port.find_hits(pile[0], pile[j], 8, 16)
            want = list(port.best_range2(q, t))
            assert rows[j][1] == len(q), (j, rows[j][1], len(q))
            assert rows[j][0] == want, (j, rows[j][0], want)
            n_window += want[4] > 0
    assert n_window > 20
    # the low-mask variant floods nothing; an unmasked run (threshold above any bucket)
    # makes hit lists that do not fit LDS and exercises the HBM scratch path
    big = _trim_ranges(engine, piles[3:], mask=100000)
    for j in range(1, len(piles[3])):
        q, t = port.find_hits(piles[3][0], piles[3][j], 8, 100000)
        assert big[0][j][1] == len(q)
        assert big[0][j][0] == list(port.best_range2(q, t)), j
    assert max(r[1] for r in big[0]) > 2048


# --------------------------------------------------------------------------
# generate_utg_consensus (src/c/falcon.c:668-773): reads laid on a
# unitig by offsets, band 500, the unitig itself as an all-match alignment, min_cov 0
# --------------------------------------------------------------------------
F7 = load_golden("f7_utg")["cases"]


@pytest.mark.parametrize("case", F7, ids=[c["name"] for c in F7])
def test_utg_consensus_golden(legacy, case):
    seq, eqv, off_after = legacy.generate_utg_consensus(case["seqs"], case["offsets"], 0, 8,
                                                        case["min_idt"])
    assert seq == case["sequence"]
    assert sha_ints(eqv) == case["eqv_sha"] and eqv[:64] == case["eqv_head"]
    assert off_after == case["offsets_after"]  # negative offsets are rewritten to 0 (:731)


def test_utg_consensus_vs_reference_build(legacy):
    """Randomised layouts against the compiled reference itself (oracle/_ref travels to the
    GPU box as a built artefact; skipped where it is absent): reads opening with an
    insertion (dropped at offset 0, hung off the base before otherwise), overhangs on
    both ends, offsets that skip the read."""
    from oracle.pyoracle import Ref, have_ref
    from falcon_amd.synth import codes_to_str, noisy
    if not have_ref():
        pytest.skip("oracle/_ref/falcon_ref.so not built here")
    ref = Ref()
    rng = np.random.default_rng(99)
    for trial in range(6):
        L = int(rng.integers(700, 5000))
        utg = rng.integers(0, 4, L).astype(np.uint8)
        seqs, offs = [codes_to_str(utg)], [0]
        for _ in range(int(rng.integers(3, 14))):
            a = int(rng.integers(-600, L + 100))
            b = a + int(rng.integers(200, 2500))
            left = rng.integers(0, 4, max(0, -a)).astype(np.uint8)
            right = rng.integers(0, 4, max(0, b - L)).astype(np.uint8)
            body = utg[max(a, 0):min(b, L)]
            rd = noisy(np.concatenate([left, body, right]), rng, float(rng.choice([0.0, 0.08, 0.15])))
            if rd.shape[0] < 20:
                continue
            seqs.append(codes_to_str(rd))
            offs.append(a)
        want = ref.generate_utg_consensus(seqs, offs, 0, 8, 0.70)
        got = legacy.generate_utg_consensus(seqs, offs, 0, 8, 0.70)
        assert got == want, trial
