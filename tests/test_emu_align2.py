"""The k_align2 kernel's SOURCE (falcon_amd/csrc/k_align2_core.h: two alignments per wavefront,
band placement and parking, the iteration tape with 1-byte cells, the trace-back) run on the
host through the lane emulator of tests/emu, against the CPU oracle: summaries, DP-cell
counts and gapped strings must be identical.  The GPU parity tests (tests/test_gpu_*.py) run
the same source compiled for gfx950; this suite is where its control logic is exercised
without a GPU -- many more shapes than a GPU box's minutes allow, and under a bounds
checker."""
import numpy as np
import pytest

from emu_driver import align_pairs
from falcon_amd.synth import codes_to_str, noisy

KEYS = ["dist", "aln_q_e", "aln_t_e", "aln_str_size", "q_aln_str", "t_aln_str", "cells"]


def _pair(rng, n, e, e2=None):
    g = rng.integers(0, 4, n, dtype=np.uint8)
    return codes_to_str(noisy(g, rng, e)), codes_to_str(noisy(g, rng, e if e2 is None else e2))


def _check(port, pairs, res, allow_handback=False, windows=None):
    n_back = 0
    for i, ((q, t), r) in enumerate(zip(pairs, res)):
        if windows:
            s1, e1, s2, e2 = windows[i]
            q, t = q[s1:e1], t[s2:e2]
        if r["err"] == 2 and allow_handback:
            n_back += 1
            continue
        assert r["err"] == 0, (i, r["err"])
        o = port.align(q, t, 150, 1)
        if not r["aligned"]:
            assert o["aln_str_size"] == 0 and o["dist"] == 0, i
            assert r["cells"] == o["cells"], i
            continue
        for k in KEYS:
            assert r[k] == o[k], (i, k, len(q), len(t))
        assert r["n_ins"] == r["n_ins_script"], i
        size = (o["aln_q_e"] + o["aln_t_e"] + o["dist"]) // 2
        assert r["accept"] == int(size > 500 and o["dist"] / size < 2.0)
    return n_back


def test_pairs_of_many_shapes(port):
    """Lengths 30..6000, divergence 0..35 %, in queue order and shuffled (different pairings in
    the wave): every alignment identical to the oracle's, both tracks used."""
    rng = np.random.default_rng(11)
    pairs, err = [], []
    for _ in range(60):
        err.append(float(rng.choice([0.0, 0.02, 0.08, 0.13, 0.2, 0.35])))
        pairs.append(_pair(rng, int(rng.integers(30, 6000)), err[-1]))
    res, st = align_pairs(pairs)
    _check(port, pairs, res, allow_handback=True)
    # what the falcon_sense regime produces (<= 13 % per read) never needs the general kernel
    assert all(r["err"] == 0 for r, e in zip(res, err) if e <= 0.13)
    assert st[0] > 0 and st[1] > 0  # paired and single iterations both occurred
    order = rng.permutation(2 * len(pairs)).astype(np.int32)
    res2, _ = align_pairs(pairs, order=order)
    _check(port, pairs, res2, allow_handback=True)
    assert all(r["err"] == 0 for r, e in zip(res2, err) if e <= 0.13)


def test_windows_inside_longer_sequences(port):
    """The falcon_sense case: (s1, e1) x (s2, e2) windows inside reads and seeds (bases before
    and behind the windows are real bases, not padding)."""
    rng = np.random.default_rng(5)
    pairs, wins = [], []
    for _ in range(24):
        n = int(rng.integers(1500, 5000))
        q, t = _pair(rng, n, 0.13)
        s1 = int(rng.integers(0, 300))  # (windows open on a k-mer match: about the same place)
        s2 = max(0, s1 + int(rng.integers(-8, 9)))
        e1, e2 = len(q) - int(rng.integers(0, 300)), len(t) - int(rng.integers(0, 300))
        pairs.append((q, t))
        wins.append((s1, e1, s2, e2))
    res, _ = align_pairs(pairs, windows=wins)
    assert _check(port, pairs, res, windows=wins) == 0


def test_long_snakes_take_the_escape_list(port):
    """Identical and nearly identical sequences: snakes of hundreds to thousands of bases do
    not fit the one-byte cell and go through the escape list."""
    rng = np.random.default_rng(3)
    pairs = []
    for n in (40, 255, 256, 300, 1000, 5000):
        g = codes_to_str(rng.integers(0, 4, n, dtype=np.uint8))
        pairs.append((g, g))
    for n in (2000, 4000, 6000):
        pairs.append(_pair(rng, n, 0.002))
        pairs.append(_pair(rng, n, 0.0, 0.01))
    res, _ = align_pairs(pairs)
    assert _check(port, pairs, res) == 0
    assert any(len(r["q_aln_str"]) > 3000 and r["dist"] < 20 for r in res)


def test_unrelated_and_unbalanced_sequences(port):
    """Rows run out (DW_banded.c:183) or one sequence ends early: unaligned summaries and DP
    cell counts match; a neighbour in the same wave is not disturbed.  (Long unrelated
    sequences open the band beyond 60 diagonals and are handed back.)"""
    rng = np.random.default_rng(8)
    pairs, short = [], []
    for _ in range(10):
        n1, n2 = (int(rng.integers(40, 100)), int(rng.integers(40, 100)))  # max_d <= 59 rows
        pairs.append((codes_to_str(rng.integers(0, 4, n1, dtype=np.uint8)),
                      codes_to_str(rng.integers(0, 4, n2, dtype=np.uint8))))
        short.append(True)
        pairs.append(_pair(rng, int(rng.integers(500, 3000)), 0.13))
        short.append(True)
    for _ in range(4):
        pairs.append((codes_to_str(rng.integers(0, 4, 1500, dtype=np.uint8)),
                      codes_to_str(rng.integers(0, 4, 1400, dtype=np.uint8))))
        short.append(False)
    q, t = _pair(rng, 3000, 0.1)
    pairs.append((q[:700], t))   # the query ends first
    pairs.append((q, t[:900]))
    pairs.append(("ACGT", "ACGT"))
    pairs.append(("A", "C"))     # max_d = 0: no row at all
    pairs.append(("ACGTACGTAC", "ACGTTCGTAC"))
    short += [True] * 5
    res, _ = align_pairs(pairs)
    _check(port, pairs, res, allow_handback=True)
    assert all(r["err"] == 0 for r, s in zip(res, short) if s)
    assert sum(1 for r in res if r["err"] == 0 and not r["aligned"]) >= 2


def test_wide_bands_park_a_track_or_hand_it_back(port):
    """High divergence widens the band: tracks take turns (parking), and a band beyond 60
    diagonals, or an alignment that does not fit a short tape ring, is handed back (err 2 --
    the engine repeats those with the general kernel) without touching its neighbour."""
    rng = np.random.default_rng(21)
    pairs = [_pair(rng, int(rng.integers(2000, 5000)), float(rng.choice([0.25, 0.3, 0.35, 0.4]))) for _ in range(16)]
    res, st = align_pairs(pairs)
    n_back = _check(port, pairs, res, allow_handback=True)
    assert st[3] > 0, "no track was ever parked"
    assert n_back == st[4]
    # a ring too short for the longer alignments
    pairs = [_pair(rng, n, 0.13) for n in (500, 3000, 600, 3200, 700, 2900, 400)]
    res, st = align_pairs(pairs, ring=1024)
    n_back = _check(port, pairs, res, allow_handback=True)
    assert n_back == 3  # (refused at the queue: max_d + 192 > ring)
    assert all(r["err"] == 0 for r, (q, _) in zip(res, pairs) if len(q) < 1000)


@pytest.mark.parametrize("seed", range(4))
def test_random_campaign(port, seed):
    """The shapes of oracle/campaign_cases.py's pair generator (band 150 cases), plus the
    bench workload's regime: 8-14 kb at 13 % error."""
    rng = np.random.default_rng(1000 + seed)
    pairs = []
    for _ in range(10):
        n = int(rng.integers(100, 4000))
        e1, e2 = float(rng.choice([0.0, 0.05, 0.13, 0.2])), float(rng.choice([0.0, 0.05, 0.13, 0.2]))
        pairs.append(_pair(rng, n, e1, e2))
    if seed == 0:
        pairs += [_pair(rng, n, 0.13) for n in (9000, 12000, 14000, 10000)]
    order = rng.permutation(2 * len(pairs)).astype(np.int32)
    res, _ = align_pairs(pairs, order=order, ring=16384)
    _check(port, pairs, res, allow_handback=True)


def test_wide_rows_run_in_the_kernel(port):
    """Bands of 61..191 diagonals (a fifth of real alignments meets one for a few rows): the
    track goes through them alone, 64 cells per pass with continuation records on the tape;
    alone in the wave it is never handed back -- aligned results, rows that run out and the
    band-too-wide exit (DW_banded.c:184) all match the oracle."""
    rng = np.random.default_rng(77)
    wide = 0
    for i in range(24):
        if i % 4 == 3:  # unrelated: the band opens until it exceeds the tolerance
            q = codes_to_str(rng.integers(0, 4, int(rng.integers(200, 2500)), dtype=np.uint8))
            t = codes_to_str(rng.integers(0, 4, int(rng.integers(200, 2500)), dtype=np.uint8))
        else:
            q, t = _pair(rng, int(rng.integers(400, 3000)), float(rng.choice([0.18, 0.2, 0.22, 0.25, 0.3])))
        res, st = align_pairs([(q, t)], ring=16384)
        assert _check(port, [(q, t)], res) == 0
        wide += int(st[6])
    assert wide > 20
    # with a neighbour waiting, a track that stays wide is handed back after A2_WIDE_PATIENCE rows
    pairs = [_pair(rng, 5000, 0.13), _pair(rng, 5000, 0.3)]
    res, st = align_pairs(pairs)
    assert res[0]["err"] == 0 and res[1]["err"] == 2
    assert _check(port, pairs, res, allow_handback=True) == 1
    res, st = align_pairs(pairs[1:])  # ... and alone it is not
    assert _check(port, pairs[1:], res) == 0


def _campaign_pairs(lo, hi):
    from oracle.campaign_cases import function_cases
    return [(q, t) for s in range(lo, hi + 1) for q, t, band in function_cases(s) if band == 150]


def _check_sub_order(port, pairs, order):
    """A queue that names only some of the pairs: those pairs, renumbered."""
    ids = sorted(set(int(g) // 2 for g in order))
    at = {p: k for k, p in enumerate(ids)}
    sub = [pairs[p] for p in ids]
    o = [2 * at[int(g) // 2] + (int(g) & 1) for g in order]
    o += [2 * k + 1 for k in range(len(sub)) if 2 * k + 1 not in set(o)]
    res, _ = align_pairs(sub, order=np.array(o, dtype=np.int32))
    _check(port, sub, res, allow_handback=True)


@pytest.mark.parametrize("case", ["wide_row_on_a_block_boundary", "cell_on_lane_0_of_an_even_row",
                                  "new_track_joins_a_full_wave", "track_out_of_rows_is_not_parked"])
def test_queue_orders_that_went_wrong_once(port, case):
    """Three wrong answers tests/emu_stress.py found in round 3, each with the queue order that
    produced it (tests/golden/emu_orders.json.gz; pairs of the frozen campaign):
    a trace-back block whose newest row was a wide row took the fast chain; a cell on lane 0
    of an even row read V[k-1] = 0 instead of 'outside the band' (the DP-cell count, not the
    path, differed); a track without rows joined a wave with fewer than four free lanes and
    read its first V[k+1] from the other track's side; (while parking moved into the row loop's
    function) a track whose rows were used up in the very iteration the pair broke up was parked
    instead of retired, and aligned on past the reference's row limit."""
    from conftest import load_golden
    c = load_golden("emu_orders")[case]
    pairs = _campaign_pairs(*c["seeds"])
    _check_sub_order(port, pairs, c["order"])


def test_random_queue_orders_over_random_tape_contents(port, monkeypatch):
    """A slice of tests/emu_stress.py in the suite: campaign pairs in random orders, the arena
    filled with random words first."""
    pairs = _campaign_pairs(208, 223)
    rng = np.random.default_rng(2083)
    for trial in range(3):
        monkeypatch.setenv("EMU_FILL", str(int(rng.integers(1, 1 << 30))))
        order = rng.permutation(2 * len(pairs)).astype(np.int32)
        res, _ = align_pairs(pairs, order=order)
        _check(port, pairs, res, allow_handback=True)


def test_the_two_primitive_layers_name_the_same_primitives():
    """k_align2_core.h is written against falcon_amd/csrc/fa_wave.h (registers and single instructions) and
    runs here against tests/emu/fa_wave.h (64-element arrays): a primitive added to one and not to the
    other would only show as a compile error of whichever build comes second -- or, for one the core does
    not use yet, not at all.  The lists must agree, but for what only the device-side code beside the
    core uses (the hand-scheduled row loop's wrapper, k_align.hip)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def names(path):
        return set(re.findall(r"\bW_FN\s+[\w:<>\s\*&]+?\b(w_\w+)\s*\(", open(path).read()))

    device = names(os.path.join(root, "falcon_amd", "csrc", "fa_wave.h"))
    emu = names(os.path.join(root, "tests", "emu", "fa_wave.h"))
    device_only = {"w_load8", "w_twice_plus",   # (k_align.hip's, not the core's)
                   "w_reduce_min"}               # (k_align2_rows.h: where a band stands when its LDS windows are filled)
    assert emu - device == set(), "the emulator implements primitives the device layer does not have"
    assert device - emu == device_only, sorted(device - emu)
    # ... and the core uses nothing outside the common list
    core = open(os.path.join(root, "falcon_amd", "csrc", "k_align2_core.h")).read()
    used = set(re.findall(r"\b(w_[a-z0-9_]+)\s*(?:<[^>]*>)?\s*\(", core))
    assert used <= (device & emu) | {"w_lds"}, sorted(used - (device & emu))


def test_the_emulator_counts_what_the_kernel_touches(port):
    """scripts/a2_bytes.py's instrument: with emu_acct_on every wave-wide load / store of the kernel's source is
    counted by what it touches (packed words, tape cells, tape records, escape list, edit scripts, alignment
    records).  The tape's layout makes three of the sums predictable: a 256-byte row of cell words per 4 iterations,
    a 16-byte record per iteration, 4 bytes of edit script per band row -- plus, per trace-back, the partial group /
    block that a2_flush_partial writes ahead of it."""
    import ctypes as C
    from emu_driver import lib
    rng = np.random.default_rng(77)
    pairs = [_pair(rng, int(n), 0.12) for n in (1500, 2200, 900, 3000, 1800, 2600)]
    lib().emu_acct_on(1)
    try:
        res, stats = align_pairs(pairs)
        tab = np.zeros((6, 2, 5))
        lib().emu_acct_get(tab.ctypes.data_as(C.c_void_p))
    finally:
        lib().emu_acct_on(0)
    _check(port, pairs, res)
    its = int(stats[0]) + int(stats[1])
    n = len(pairs)
    rows = sum(r["dist"] + 1 for r in res if r["aligned"])
    cells_w, recs_w, script_w = tab[1][1][1], tab[2][1][1], tab[4][1][1]
    assert its // 4 * 256 <= cells_w <= (its // 4 + n + 1) * 256, (its, cells_w)
    assert its // 64 * 1024 <= recs_w <= its * 16 + n * 1024, (its, recs_w)
    assert script_w == 4 * rows, (rows, script_w)
    # the trace-back reads every block of 64 iterations it walks as 16-byte records, and one cell word per row
    assert tab[2][0][1] >= 16 * rows and tab[1][0][1] == 4 * rows, (rows, tab[2][0], tab[1][0])
    assert tab[3].sum() == 0   # (no snake of >= 255 bases here: the escape list is not touched)
