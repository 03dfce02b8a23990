"""bench.py --workload trim | align1500 | utg: throughput lines for the paths SURVEY.md 8(f) lists beside the
falcon_sense hot path.  Each leg keeps bench.py's contract -- W warm-up steps, K timed steps bracketed by device
syncs, one JSON line with `roofline` and `cpu_baseline` -- on its own metric (these are not BASELINE.json's
headline, and say so), with the algorithmic bytes of DESIGN.md section 5a and the compiled reference
(oracle/_ref, the checker, timed on a bounded sample of the very same input) beside it.

  trim       `--trim`: find_best_aln_range2 of every read on its seed (falcon_kit/mains/consensus.py:123-158,
             src/c/kmer_lookup.c:195-204, :429-585) -- fa_batch_trim_windows over a resident batch of piles
  align1500  DWA.align at band_tolerance 1500 on contig-sized pairs (falcon_kit/mains/graph_to_contig.py:52-105,
             src/c/DW_banded.c:115-330) -- fa_align_pairs -> k_align_wide
  utg        generate_utg_consensus (src/c/falcon.c:668-773): reads laid on a unitig by offsets -- fa_utg_consensus
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0
SO = os.path.join(ROOT, "falcon_amd", "libfalcon_amd.so")


def _sync():
    """(every call timed here returns with its results on the host: nothing is left in flight)"""


def _timed(step, warmup, steps):
    for _ in range(warmup):
        step()
    _sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _sync()
    return time.perf_counter() - t0


def _line(metric, unit, value, steps, warmup, elapsed, dtype, workload, extra_cfg, kernel, alg_bytes, cpu, more):
    ms = elapsed / max(1, steps) * 1e3
    ach = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    res = {"metric": metric, "value": round(value, 1), "unit": unit, "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": dtype, "data": "synthetic",
           "config": dict({"workload": workload}, **extra_cfg),
           "not_the_headline": "SURVEY.md 8(f) path; BASELINE.json's metric is the default workload's",
           "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": int(alg_bytes),
                        "avg_launch_ms": round(ms, 4),
                        "avg_launch_ms_is": "host wall time of the whole call (its kernels, the download of its results)",
                        "traffic": None},
           "cpu_baseline": cpu}
    res.update(more)
    return res


# ------------------------------------------------------------------------------------------------
def _trim(args):
    from benchlib.workloads import WORKLOADS, gen_piles
    from falcon_amd.engine import Engine
    wl = WORKLOADS["ecoli"]
    n = args.piles if args.piles_given else 1024
    piles = gen_piles([1000003 + i for i in range(n)], max(1, min(32, (os.cpu_count() or 1))), wl)
    eng = Engine(0)
    b = eng.batch(piles)
    elapsed = _timed(lambda: b.trim_windows(8, 16), args.warmup, args.steps)
    n_reads = sum(len(p) - 1 for p in piles)
    L = sum(len(s) for p in piles for s in p)
    T = sum(len(p[0]) for p in piles)
    # every base read once packed (L/4), the seed index built (T/4 + 8T + 2 x 4 x 65537 per pile),
    # one 32-byte window record per read
    alg = L // 4 + T // 4 + 8 * T + 2 * 4 * 65537 * len(piles) + 32 * n_reads
    # parity + CPU figure: the compiled reference on the first piles, one lookup per seed like get_consensus_with_trim
    got = []
    g = 0
    for p in piles:
        rows = []
        for _ in p:
            r = b.range(g)
            rows.append((r["s1"], r["e1"], r["s2"], r["e2"], r["score"]))
            g += 1
        got.append(rows)
    b.free()
    eng.close()
    cpu = None
    mism = 0
    try:
        import ctypes as C
        from oracle.pyoracle import Ref, _KmerMatch  # noqa: F401
        ref = Ref()
        lib = ref.lib
        t_cpu, n_cpu, k_piles = 0.0, 0, 0
        for pi, p in enumerate(piles):
            if t_cpu > 12.0:
                break
            seed = p[0].encode() if isinstance(p[0], str) else p[0]
            t0 = time.perf_counter()
            lk = lib.allocate_kmer_lookup(1 << 16)
            sa = lib.allocate_seq(len(seed))
            sda = lib.allocate_seq_addr(len(seed))
            lib.add_sequence(0, 8, seed, len(seed), sda, sa, lk)
            lib.mask_k_mer(1 << 16, lk, 16)
            rows = []
            for q in p[1:]:
                q = q.encode() if isinstance(q, str) else q
                km = lib.find_kmer_pos_for_seq(q, len(q), 8, sda, lk)
                if km[0].count:
                    r = lib.find_best_aln_range2(km, 8, 400, 25)
                    rows.append((r[0].s1, r[0].e1, r[0].s2, r[0].e2, r[0].score))
                    lib.free_aln_range(r)
                else:
                    rows.append((0, 0, 0, 0, 0))
                lib.free_kmer_match(km)
            lib.free_seq_addr_array(sda); lib.free_seq_array(sa); lib.free_kmer_lookup(lk)
            t_cpu += time.perf_counter() - t0
            n_cpu += len(rows)
            k_piles += 1
            mism += sum(1 for a, bb in zip(rows, got[pi][1:]) if tuple(a) != tuple(bb))
        cpu = {"value": round(n_cpu / t_cpu, 1), "unit": "reads/s", "cores": 1, "kind": "reference",
               "sample": "find_kmer_pos_for_seq + find_best_aln_range2 of every read of the first %d piles on one lookup per "
                         "seed (mask 16), compiled reference through ctypes, one thread" % k_piles,
               "parity_checked_reads": n_cpu, "parity_mismatches": mism}
    except Exception as e:  # the reference build is a checker: without it the line still stands
        cpu = {"value": None, "unit": "reads/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
    return _line("trim_windows_reads_per_sec", "reads/s", n_reads * args.steps / elapsed, args.steps, args.warmup, elapsed,
                 "int32 (8-mer hits, diagonal bins, running score)",
                 "--trim windows (consensus.py:123-158) of %d E. coli-like piles: 20 kb seeds x 40x, mask 16" % len(piles),
                 {"piles_per_step": len(piles), "reads_per_step": n_reads}, "k_seed_index + k_trimwin", alg, cpu,
                 {"piles_per_sec": round(len(piles) * args.steps / elapsed, 1)})


# ------------------------------------------------------------------------------------------------
def _align1500(args):
    from falcon_amd.engine import Engine
    from falcon_amd.synth import codes_to_str, noisy
    rng = np.random.default_rng(5)
    n_pairs = args.piles if args.piles_given else 16
    pairs = []
    for i in range(n_pairs):
        n = (60000, 100000, 150000, 230000)[i % 4]
        e = (0.02, 0.05, 0.01, 0.03)[i % 4]
        t = rng.integers(0, 4, n).astype(np.uint8)
        pairs.append((codes_to_str(noisy(t, rng, e)), codes_to_str(t)))
    eng = Engine(0)
    res = [None]

    def step():
        res[0] = eng.align_pairs(pairs, band=1500, want_str=True)
    elapsed = _timed(step, args.warmup, args.steps)
    eng.close()
    cols = sum(r["aln_str_size"] for r in res[0])
    D = sum(r["dist"] + 1 for r in res[0])
    L = sum(len(q) + len(t) for q, t in pairs)
    # both sequences read packed (L/4), 8 bytes per band row (the O(ND) front's bookkeeping), the two
    # gapped strings written (2 bytes per column); the cells of the band are not counted by this entry point
    alg = L // 4 + 8 * D + 2 * cols
    cpu, mism = None, 0
    try:
        from oracle.pyoracle import Ref
        ref = Ref()
        t_cpu, n_done, c_done = 0.0, 0, 0
        for (q, t), r in zip(pairs, res[0]):
            if t_cpu > 15.0:
                break
            t0 = time.perf_counter()
            o = ref.align(q, t, 1500, 1)
            t_cpu += time.perf_counter() - t0
            n_done += 1
            c_done += o["aln_str_size"]
            mism += int(any(o[k] != r[k] for k in ("dist", "aln_str_size", "aln_q_e", "aln_t_e", "q_aln_str", "t_aln_str")))
        cpu = {"value": round(c_done / t_cpu, 1), "unit": "aligned columns/s", "cores": 1, "kind": "reference",
               "sample": "align(q, t, 1500, 1) of the first %d pairs, compiled reference, one thread" % n_done,
               "parity_checked_pairs": n_done, "parity_mismatches": mism}
    except Exception as e:
        cpu = {"value": None, "unit": "aligned columns/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
    return _line("align_band1500_columns_per_sec", "aligned columns/s", cols * args.steps / elapsed, args.steps, args.warmup,
                 elapsed, "int32 (furthest-reaching x per diagonal)",
                 "DWA.align at band_tolerance 1500 (graph_to_contig.py:52-105): %d pairs of 60-230 kb, 1-5 %% apart, "
                 "one fa_align_pairs call" % n_pairs, {"pairs_per_step": n_pairs, "columns_per_step": cols, "rows_per_step": D},
                 "k_align_wide", alg, cpu, {"pairs_per_sec": round(n_pairs * args.steps / elapsed, 2)})


# ------------------------------------------------------------------------------------------------
def _utg(args):
    from falcon_amd.synth import codes_to_str, noisy
    from oracle.pyoracle import LegacyABI   # (only its ctypes prototypes: drives the PRODUCT library)
    rng = np.random.default_rng(707)
    U = 90000 if not args.piles_given else max(5000, min(99000, args.piles))   # (targets < 100 000 bases: falcon.c:343)
    utg = rng.integers(0, 4, U).astype(np.uint8)
    seqs, offs = [codes_to_str(utg)], [0]
    cover, read_len = 30, 12000
    for a in range(-read_len // 2, U, max(1, read_len // cover)):
        b = a + int(rng.integers(read_len // 2, read_len))
        left = rng.integers(0, 4, max(0, -a)).astype(np.uint8)
        right = rng.integers(0, 4, max(0, b - U)).astype(np.uint8)
        rd = noisy(np.concatenate([left, utg[max(a, 0):min(b, U)], right]), rng, 0.12)
        seqs.append(codes_to_str(rd))
        offs.append(int(a))
    prod = LegacyABI(SO)
    out = [None]

    def step():
        out[0] = prod.generate_utg_consensus(seqs, offs, 0, 8, 0.70)
    elapsed = _timed(step, args.warmup, args.steps)
    cns = out[0][0]
    L = sum(len(s) for s in seqs)
    A = L  # columns of the accepted alignments ~ the reads' bases
    # the falcon_sense formula without the seed index: L/4 + 16 A + 12 T + 5 O
    alg = L // 4 + 16 * A + 12 * U + 5 * len(cns)
    cpu = None
    try:
        from oracle.pyoracle import Ref
        ref = Ref()
        t0 = time.perf_counter()
        want = ref.generate_utg_consensus(seqs, offs, 0, 8, 0.70)
        t_cpu = time.perf_counter() - t0
        cpu = {"value": round(len(want[0]) / t_cpu, 1), "unit": "consensus bases/s", "cores": 1, "kind": "reference",
               "sample": "the same unitig and reads, one call of the compiled reference's generate_utg_consensus",
               "parity_checked_unitigs": 1, "parity_mismatches": int(want[0] != cns or want[1] != out[0][1])}
    except Exception as e:
        cpu = {"value": None, "unit": "consensus bases/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
    return _line("utg_consensus_bases_per_sec", "consensus bases/s", len(cns) * args.steps / elapsed, args.steps, args.warmup,
                 elapsed, "u8/int32 (2-bit packed bases, integer DP)",
                 "generate_utg_consensus (falcon.c:668-773): one unitig of %d bases, %d reads of 6-12 kb laid on it by "
                 "offset (band 500), 12 %% apart, one call through the legacy symbol" % (U, len(seqs) - 1),
                 {"unitig_bases": U, "reads": len(seqs) - 1}, "k_align (band 500) + k_tags + k_links2 + k_score1 + k_backtrace",
                 alg, cpu, {})


LEGS = {"trim": _trim, "align1500": _align1500, "utg": _utg}


def run(args, out=None):
    out = sys.stdout if out is None else out
    res = LEGS[args.workload](args)
    out.write(json.dumps(res) + "\n")
    out.flush()
    return res
