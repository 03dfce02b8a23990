#!/usr/bin/env python3
"""Generate tests/golden/*.json.gz from the REAL reference.  TEST INFRASTRUCTURE ONLY.

Runs only in the dev container (needs /root/reference and oracle/_ref/falcon_ref.so,
built by `make -C oracle`).  Outputs are data: inputs + the reference's answers.

Sources of truth used here
  * function and pile level (F1-F4, F6): the compiled reference C
    (/root/reference/src/c/{kmer_lookup,DW_banded,falcon}.c) through its own C ABI;
  * CLI level (F5): the reference's own Python driver
    /root/reference/falcon_kit/mains/consensus.py (run(), get_seq_data(),
    get_longest_reads(), format_seq() unmodified), imported under python3 with
      - a stub module `ext_falcon` whose __file__ is the compiled reference .so
        (what falcon_kit/falcon_kit.py:11,45 expects), and
      - get_consensus_without_trim / get_consensus_with_trim wrapped so that the
        sequences cross the ctypes boundary as bytes (python3's c_char_p rejects
        str, consensus.py:110); the wrapped functions themselves are the
        reference's.
The reference's test-suite holds no vectors for this path (test/test_consensus.py:5-9
is --help only); test_data/t1.fa is the only reference data file used (config 1).
"""
from __future__ import annotations

import contextlib
import gzip
import hashlib
import io
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

from falcon_amd.synth import (codes_to_str, make_pile, noisy, pile_to_la4falcon,  # noqa: E402
                              pile_to_seqs)
from oracle.pyoracle import REF_SO, Ref  # noqa: E402

REFERENCE = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def dump(name, obj):
    path = os.path.join(OUT, name + ".json.gz")
    raw = json.dumps(obj, separators=(",", ":"), sort_keys=True).encode()
    with open(path, "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
            g.write(raw)
    print("%-28s %8d bytes (%d raw)" % (name, os.path.getsize(path), len(raw)))


def sha_ints(xs):
    return hashlib.sha1(np.asarray(xs, dtype="<i4").tobytes()).hexdigest()


def rand_seq(rng, n):
    return codes_to_str(rng.integers(0, 4, n, dtype=np.uint8))


def str_codes(s):
    return np.frombuffer(s.encode(), dtype=np.uint8).copy()


def to_codes(s):
    lut = np.zeros(256, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    return lut[np.frombuffer(s.encode(), dtype=np.uint8)]


def read_t1():
    with open(os.path.join(REFERENCE, "test_data", "t1.fa")) as f:
        lines = f.read().split("\n")
    return lines[0][1:], "".join(lines[1:]).strip()


# --------------------------------------------------------------------------
def gen_f1_f2(R):
    rng = np.random.default_rng(101)
    cases = []

    def add(name, seed, query, mask=-1):
        q, t = R.find_hits(seed, query, 8, mask)
        c = dict(name=name, seed=seed, query=query, K=8, mask=mask, count=len(q),
                 sha_q=sha_ints(q), sha_t=sha_ints(t))
        if len(q) <= 400:
            c["q"], c["t"] = q, t
        c["range_48_5"] = list(R.best_range(q, t, 48, 5))
        c["range_80_50"] = list(R.best_range(q, t, 80, 50))  # falcon_kit.py:140
        if mask >= 0 and len(q):
            c["range2"] = list(R.best_range2(q, t))
        cases.append(c)

    g = rng.integers(0, 4, 600, dtype=np.uint8)
    add("identical_600", codes_to_str(g), codes_to_str(g))
    add("noisy_600", codes_to_str(noisy(g, rng, 0.13)), codes_to_str(noisy(g, rng, 0.13)))
    add("unrelated_600", rand_seq(rng, 600), rand_seq(rng, 600))
    add("short_query_9", codes_to_str(g), codes_to_str(g[100:109]))
    add("query_len_eq_K_plus1", codes_to_str(g), codes_to_str(g[7:16]))
    rep = "ACGTTGCA" * 40
    add("tandem_repeat", rep + rand_seq(rng, 100), rep[:200])
    add("homopolymer", "A" * 300 + rand_seq(rng, 200), "A" * 100 + rand_seq(rng, 50))
    g2 = rng.integers(0, 4, 6000, dtype=np.uint8)
    s2, q2 = codes_to_str(noisy(g2, rng, 0.13)), codes_to_str(noisy(g2[1500:5200], rng, 0.13))
    add("noisy_6000_window", s2, q2)
    add("noisy_6000_window_masked", s2, q2, mask=16)
    add("tandem_repeat_masked", rep + rand_seq(rng, 2000), rep[:200] + rand_seq(rng, 300), mask=16)
    g3 = rng.integers(0, 4, 9000, dtype=np.uint8)
    add("noisy_9000_masked", codes_to_str(noisy(g3, rng, 0.12)),
        codes_to_str(noisy(g3[200:8800], rng, 0.12)), mask=16)
    # a query that is two far-apart seed windows glued together (two diagonals)
    add("chimeric", s2, codes_to_str(noisy(g2[300:1500], rng, 0.1)) +
        codes_to_str(noisy(g2[4000:5500], rng, 0.1)))
    dump("f1_f2_hits_ranges", dict(cases=cases))

    # F2 extra: hand-made hit lists
    extra = []

    def addh(name, q, t, b, th):
        extra.append(dict(name=name, q=q, t=t, bin=b, th=th, range=list(R.best_range(q, t, b, th))))

    addh("single_hit", [0], [10], 48, 5)
    addh("below_threshold", [0, 4, 8, 12], [0, 4, 8, 12], 48, 5)
    addh("exactly_threshold_plus1", [0, 4, 8, 12, 16, 20], [0, 4, 8, 12, 16, 20], 48, 5)
    addh("gap_reset", [0, 4, 8, 12, 16, 20, 400, 404, 408, 412, 416, 420, 424],
         [0, 4, 8, 12, 16, 20, 400, 404, 408, 412, 416, 420, 424], 48, 5)
    addh("repeated_query_pos", [0, 0, 0, 4, 4, 8, 8, 8, 12, 16, 20, 24],
         [0, 1, 2, 4, 5, 8, 9, 10, 12, 16, 20, 24], 48, 5)
    addh("tie_between_bins", [0, 4, 8, 12, 16, 20, 1000, 1004, 1008, 1012, 1016, 1020],
         [500, 504, 508, 512, 516, 520, 0, 4, 8, 12, 16, 20], 48, 5)
    dump("f2_ranges_extra", dict(cases=extra))


def gen_f3(R):
    rng = np.random.default_rng(303)
    cases = []

    def add(name, q, t, band=150, want=1):
        a = R.align(q, t, band, want)
        cases.append(dict(name=name, q=q, t=t, band=band, want_str=want, expect=a))

    g = rng.integers(0, 4, 700, dtype=np.uint8)
    s = codes_to_str(g)
    add("identical_700", s, s)
    add("identical_nostr", s, s, want=0)
    add("noisy_700", codes_to_str(noisy(g, rng, 0.13)), codes_to_str(noisy(g, rng, 0.13)))
    add("noisy_700_nostr", codes_to_str(noisy(g, rng, 0.13)), codes_to_str(noisy(g, rng, 0.13)),
        want=0)
    add("q_prefix_of_t", s[:300], s)
    add("t_prefix_of_q", s, s[:300])
    add("unrelated_maxd", rand_seq(rng, 400), rand_seq(rng, 400))
    add("tiny_1x1_match", "A", "A")
    add("tiny_1x1_mismatch", "A", "C")
    add("tiny_3x4", "ACG", "ACGT")
    add("tiny_5x5_sub", "ACGTA", "ACCTA")
    add("len10_one_ins", "ACGTACGTAC", "ACGTAACGTAC")
    # band break: a large indel forces the band past 2*tolerance
    g2 = rng.integers(0, 4, 3000, dtype=np.uint8)
    add("band20_break", codes_to_str(noisy(g2, rng, 0.2)), codes_to_str(noisy(g2, rng, 0.2)), band=20)
    add("big_deletion", codes_to_str(np.concatenate([g2[:1000], g2[1400:]])), codes_to_str(g2))
    add("noisy_3000", codes_to_str(noisy(g2, rng, 0.13)), codes_to_str(noisy(g2, rng, 0.13)))
    add("noisy_3000_band1500", codes_to_str(noisy(g2, rng, 0.13)),
        codes_to_str(noisy(g2, rng, 0.13)), band=1500)
    add("noisy_3000_30pct", codes_to_str(noisy(g2, rng, 0.30)), codes_to_str(noisy(g2, rng, 0.30)))
    add("homopolymer_slip", "A" * 200 + s[:200], "A" * 230 + s[:200])
    g3 = rng.integers(0, 4, 12000, dtype=np.uint8)
    add("noisy_12000", codes_to_str(noisy(g3, rng, 0.13)), codes_to_str(noisy(g3, rng, 0.13)))
    dump("f3_align", dict(cases=cases))


def gen_f4(R):
    rng = np.random.default_rng(404)
    cases = []

    def add(name, seqs, min_cov=4, min_idt=0.70):
        seq, eqv = R.generate_consensus(seqs, min_cov, 8, min_idt)
        cases.append(dict(name=name, seqs=seqs, min_cov=min_cov, K=8, min_idt=min_idt,
                          sequence=seq, eqv_sha=sha_ints(eqv), eqv_head=eqv[:64]))
        print("   pile %-24s n_seq=%3d seed=%6d -> cns %6d" % (name, len(seqs), len(seqs[0]), len(seq)))

    s2000 = rand_seq(rng, 2000)
    add("identical_copies_12", [s2000] * 12)                       # Q1-Q3
    add("identical_copies_3_lowcov", [s2000] * 3, min_cov=4)       # lower-case output (Q7)
    seed, reads = make_pile(11, S=5000, coverage=15, min_read=800, mean_read=3000, sd_read=1000)
    add("noisy_5k_x15", [codes_to_str(x) for x in pile_to_seqs(seed, reads)])
    add("noisy_5k_x15_mincov0_idt85", [codes_to_str(x) for x in pile_to_seqs(seed, reads)],
        min_cov=0, min_idt=0.85)
    seed, reads = make_pile(12, S=5000, coverage=20, het=0.01, min_read=800, mean_read=3000,
                            sd_read=1000)
    add("het_5k_x20", [codes_to_str(x) for x in pile_to_seqs(seed, reads)])
    seed, reads = make_pile(13, S=4000, coverage=10, min_read=800, mean_read=2500, sd_read=800)
    seqs = [codes_to_str(x) for x in pile_to_seqs(seed, reads)]
    junk = [rand_seq(rng, 3000), rand_seq(rng, 1200), seqs[2][:90], seqs[3][:450]]
    add("with_rejected_reads", seqs[:3] + junk[:2] + seqs[3:8] + junk[2:] + seqs[8:])
    add("no_usable_reads", [s2000, rand_seq(rng, 2500), rand_seq(rng, 1800)])
    add("seed_only", [s2000])
    add("seed_plus_self", [s2000, s2000])
    # config 1: test_data/t1.fa-derived tiny pile
    _t1_id, t1 = read_t1()
    r1 = np.random.default_rng(1)
    t1c = to_codes(t1)
    derived = [codes_to_str(noisy(t1c, r1, 0.12)) for _ in range(20)]
    add("t1_config1", [codes_to_str(x) for x in
                       pile_to_seqs(t1c, [to_codes(d) for d in derived])])
    dump("f4_piles", dict(cases=cases))


# --------------------------------------------------------------------------
def import_reference_driver():
    stub = types.ModuleType("ext_falcon")
    stub.__file__ = REF_SO
    sys.modules["ext_falcon"] = stub
    sys.path.insert(0, REFERENCE)
    import falcon_kit.mains.consensus as mod  # the reference's own driver
    import falcon_kit.multiproc  # noqa: F401
    for fn_name in ("get_consensus_without_trim", "get_consensus_with_trim"):
        orig = getattr(mod, fn_name)
        if getattr(orig, "_marshalling_wrapped", False):  # (imported before in this process)
            continue

        def wrapped(c_input, _orig=orig):
            seqs, seed_id, config = c_input
            cns, sid = _orig(([s.encode("ascii") for s in seqs], seed_id, config))
            return cns.decode("ascii"), sid
        wrapped._marshalling_wrapped = True
        setattr(mod, fn_name, wrapped)
    return mod


def run_reference_cli(mod, argv, stdin_text):
    old_stdin = sys.stdin
    sys.stdin = io.StringIO(stdin_text)
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            mod.main(["fc_consensus"] + argv + ["--n-core", "0"])
    finally:
        sys.stdin = old_stdin
    return buf.getvalue()


def gen_f5_f6(R):
    mod = import_reference_driver()
    rng = np.random.default_rng(505)
    text = []
    nid = 1
    # pile A: regular
    seed, reads = make_pile(21, S=4000, coverage=14, min_read=800, mean_read=2500, sd_read=800)
    text.append(pile_to_la4falcon("00000100", seed, reads, nid)); nid += len(reads)
    # pile B: discarded with '* *'
    seed, reads = make_pile(22, S=2000, coverage=12, min_read=500, mean_read=1200, sd_read=300)
    text.append(pile_to_la4falcon("00000200", seed, reads, nid).replace("+ +", "* *")); nid += len(reads)
    # pile C: too few reads (< min_n_read)
    seed, reads = make_pile(23, S=2000, coverage=12, min_read=500, mean_read=1200, sd_read=300)
    text.append(pile_to_la4falcon("00000300", seed, reads[:5], nid)); nid += 5
    # pile D: low coverage (read_cov // seed_len < min_cov_aln)
    seed, reads = make_pile(24, S=4000, coverage=6, min_read=500, mean_read=1500, sd_read=300)
    text.append(pile_to_la4falcon("00000400", seed, reads, nid)); nid += len(reads)
    # pile E: duplicate ids, a junk 3-token line, a short consensus region split by low coverage
    seed, reads = make_pile(25, S=5000, coverage=16, min_read=800, mean_read=1500, sd_read=300)
    t = pile_to_la4falcon("00000500", seed, reads, nid); nid += len(reads)
    lines = t.split("\n")
    lines.insert(3, lines[2])                     # duplicated read id -> ignored
    lines.insert(5, "garbage line with tokens")   # != 2 tokens -> ignored
    text.append("\n".join(lines))
    # pile F: two coverage islands -> several [ACGT]+ regions under --min-cov 4
    g = rng.integers(0, 4, 6000, dtype=np.uint8)
    seedF = noisy(g, rng, 0.10)
    readsF = []
    for lo, hi, n in ((0, 2500, 14), (3500, 6000, 14), (2300, 3700, 3)):
        for _ in range(n):
            readsF.append(noisy(g[lo:hi], rng, 0.12))
    text.append(pile_to_la4falcon("00000600", seedF, readsF, nid)); nid += len(readsF)
    stdin_text = "".join(text) + "- -\nAFTER_EOF ACGT\n"

    runs = []
    for argv in (
        [],
        ["--output-multi", "--min-idt", "0.70", "--min-cov", "4", "--max-n-read", "200"],
        ["--output-full", "--min-cov", "4"],
        ["--output-multi", "--min-cov", "4", "--max-n-read", "12"],
        ["--min-cov", "4", "--min-len-aln", "1000", "--max-cov-aln", "8"],
        ["--output-multi", "--min-cov", "4", "--min-n-read", "4", "--min-cov-aln", "2"],
    ):
        out = run_reference_cli(mod, argv, stdin_text)
        runs.append(dict(argv=argv, stdout=out))
        print("   cli %-70s -> %d bytes, %d records" % (" ".join(argv), len(out), out.count(">")))
    dump("f5_cli", dict(stdin=stdin_text, runs=runs))

    # F6: --trim
    trim_runs = []
    for argv in (["--trim", "--output-multi", "--min-cov", "4"],
                 ["--trim", "--min-cov", "4", "--trim-size", "20", "--edge-tolerance", "300"]):
        out = run_reference_cli(mod, argv, stdin_text)
        trim_runs.append(dict(argv=argv, stdout=out))
        print("   cli %-70s -> %d bytes, %d records" % (" ".join(argv), len(out), out.count(">")))
    dump("f6_cli_trim", dict(stdin_from="f5_cli", runs=trim_runs))


def gen_f7(R):
    """generate_utg_consensus (src/c/falcon.c:668-773): reads laid on a unitig by offsets."""
    rng = np.random.default_rng(707)
    cases = []

    def add(name, seqs, offsets, min_idt=0.70):
        seq, eqv, off_after = R.generate_utg_consensus(seqs, offsets, 0, 8, min_idt)
        cases.append(dict(name=name, seqs=seqs, offsets=list(offsets), min_idt=min_idt,
                          sequence=seq, eqv_sha=sha_ints(eqv), eqv_head=eqv[:64],
                          offsets_after=off_after))
        print("   utg  %-24s n_seq=%3d utg=%6d -> cns %6d" % (name, len(seqs), len(seqs[0]), len(seq)))

    def layout(utg_codes, spans, e):
        """reads = noisy copies of utg[a:b] (a may be negative / b beyond the end: the
        overhang is random sequence); offset = a."""
        seqs, offs = [codes_to_str(utg_codes)], [0]
        L = utg_codes.shape[0]
        for a, b in spans:
            left = rng.integers(0, 4, max(0, -a)).astype(np.uint8)
            right = rng.integers(0, 4, max(0, b - L)).astype(np.uint8)
            body = utg_codes[max(a, 0):min(b, L)]
            rd = noisy(np.concatenate([left, body, right]), rng, e)
            seqs.append(codes_to_str(rd))
            offs.append(int(a))
        return seqs, offs

    utg = rng.integers(0, 4, 6000).astype(np.uint8)
    spans = [(0, 2500), (300, 3300), (1200, 4000), (2500, 6000), (3500, 6400), (-400, 2200),
             (-900, 900), (5950, 7500), (-3000, 100), (4000, 5200), (100, 5900), (1800, 2500)]
    add("utg6k_mixed_offsets", *layout(utg, spans, 0.10))
    add("utg6k_clean_reads", *layout(utg, spans[:6], 0.0))
    add("utg6k_noisy_idt85", *layout(utg, spans, 0.14), min_idt=0.85)
    add("utg_alone", [codes_to_str(utg)], [0])
    small = rng.integers(0, 4, 420).astype(np.uint8)      # shorter than the 500-column rule
    add("utg420_short", *layout(small, [(0, 420), (-50, 400), (30, 420)], 0.05))
    long_utg = rng.integers(0, 4, 15000).astype(np.uint8)
    add("utg15k_longer_than_reads", *layout(long_utg, [(i, i + 4000) for i in range(-1000, 14000, 900)], 0.12))
    dump("f7_utg", dict(cases=cases))


# the full-size piles of SURVEY.md 8d: config 2/3 (E. coli-like), 4 (dmel-like: the
# max_n_read = 200 cap binds), 5 (Arabidopsis-like: two haplotypes 0.5 % apart)
F8_CONFIGS = [
    dict(name="config2_ecoli_like", seed=7, S=20000, coverage=40.0, het=0.0),
    dict(name="config4_dmel_like", seed=11, S=30000, coverage=80.0, het=0.0),
    dict(name="config5_arabidopsis_like", seed=31, S=25000, coverage=60.0, het=0.005),
]


def f8_pile(cfg):
    seed, reads = make_pile(cfg["seed"], S=cfg["S"], coverage=cfg["coverage"], het=cfg["het"])
    return [codes_to_str(x) for x in pile_to_seqs(seed, reads, 200)]


def gen_f8(R):
    """One canonical pile per benchmark configuration at full size.  The inputs are a
    pure function of the generator parameters (falcon_amd/synth.py), so the fixture
    holds the parameters, a digest of the input and the reference's answer."""
    cases = []
    for cfg in F8_CONFIGS:
        seqs = f8_pile(cfg)
        seq, eqv = R.generate_consensus(seqs, 4, 8, 0.70)
        cases.append(dict(cfg, n_seq=len(seqs), n_bases=sum(map(len, seqs)),
                          input_sha=hashlib.sha1("\n".join(seqs).encode()).hexdigest(),
                          min_cov=4, K=8, min_idt=0.70, sequence=seq, eqv_sha=sha_ints(eqv)))
        print("   pile %-26s n_seq=%3d bases=%8d seed=%6d -> cns %6d"
              % (cfg["name"], len(seqs), sum(map(len, seqs)), len(seqs[0]), len(seq)))
    dump("f8_configs", dict(cases=cases))


def _f10_one(seed):
    """(worker process: the reference C is not re-entrant, one Ref per process)"""
    s, rd = make_pile(seed, S=20000, coverage=40.0)
    seqs = [codes_to_str(x) for x in pile_to_seqs(s, rd, 200)]
    seq, eqv = Ref().generate_consensus(seqs, 4, 8, 0.70)
    return dict(seed=seed, n_seq=len(seqs), input_sha=hashlib.sha1("\n".join(seqs).encode()).hexdigest()[:16],
                cns_len=len(seq), cns_sha=hashlib.sha1(seq.encode()).hexdigest(), eqv_sha=sha_ints(eqv))


def gen_f10(R):
    """The 72 bench-scale piles of tests/test_gpu_parity.py::test_bench_scale_batch_properties
    (BASELINE config 2: 20 kb seeds x 40x, e = 0.13; seeds 7000..7071) through the compiled
    reference: per pile a digest of the input, of the consensus string and of eqv."""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(16) as pool:
        cases = pool.map(_f10_one, range(7000, 7072))
    dump("f10_bench72", dict(S=20000, coverage=40.0, min_cov=4, K=8, min_idt=0.70, cases=cases))


def gen_f11(R):
    """BASELINE config 1 at the CLI level: the test_data/t1.fa-derived pile of f4's
    `t1_config1` as LA4Falcon text through the reference's OWN driver (its main() under
    python3, as for f5), in the three output modes."""
    mod = import_reference_driver()
    t1_id, t1 = read_t1()
    r1 = np.random.default_rng(1)
    t1c = to_codes(t1)
    derived = [noisy(t1c, r1, 0.12) for _ in range(20)]
    stdin_text = pile_to_la4falcon(t1_id, t1c, derived, 1) + "- -\n"
    runs = []
    for argv in ([], ["--output-multi"], ["--output-full"], ["--output-multi", "--min-cov", "6", "--min-idt", "0.75"]):
        out = run_reference_cli(mod, argv, stdin_text)
        runs.append(dict(argv=argv, stdout=out))
        print("   cli %-60s -> %d bytes, %d records" % (" ".join(argv), len(out), out.count(">")))
    dump("f11_cli_config1", dict(stdin_sha=hashlib.sha1(stdin_text.encode()).hexdigest(), seed_id=t1_id,
                                 n_reads=len(derived), runs=runs))


def main():
    os.makedirs(OUT, exist_ok=True)
    R = Ref()
    if len(sys.argv) > 1 and sys.argv[1] in ("f7", "f8", "f10", "f11"):  # (the other fixtures are already committed)
        {"f7": gen_f7, "f8": gen_f8, "f10": gen_f10, "f11": gen_f11}[sys.argv[1]](R)
        return
    gen_f1_f2(R)
    gen_f3(R)
    gen_f4(R)
    gen_f5_f6(R)
    gen_f7(R)
    gen_f8(R)


if __name__ == "__main__":
    main()
