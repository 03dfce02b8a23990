#!/usr/bin/env python3
"""Timings of the secondary entry points, on the GPU box:
  * `align` at band_tolerance 1500 on 50-250 kb pairs (graph_to_contig.py:52-105) -- k_align_wide
  * --trim windows of a bench-sized batch (fa_batch_trim_windows: k_seed_index + k_trimwin)
  * route (b) of INTEGRATION.md: the legacy one-pile-per-call generate_consensus symbol
    (a batch of one behind the library's global lock) against the batch ABI on the same piles
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from falcon_amd.engine import Engine  # noqa: E402
from falcon_amd.synth import codes_to_str, make_pile, noisy, pile_to_seqs  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(5)
print("== align, band 1500 (k_align_wide), one pair per call; second call of each (buffers exist)")
for n, e in ((50000, 0.10), (50000, 0.02), (150000, 0.05), (250000, 0.02), (250000, 0.003)):
    t = rng.integers(0, 4, n).astype(np.uint8)
    pair = [(codes_to_str(noisy(t, rng, e)), codes_to_str(t))]
    eng.align_pairs(pair, band=1500, want_str=True)
    t0 = time.perf_counter()
    (r,) = eng.align_pairs(pair, band=1500, want_str=True)
    dt = time.perf_counter() - t0
    print("  %6d bp at %4.1f %% divergence: dist %6d, %7d columns, %.3f s" % (n, 100 * e, r["dist"], r["aln_str_size"], dt))

piles = []
for i in range(256):
    s, rd = make_pile(1000003 + i, S=20000, coverage=40.0)
    piles.append([codes_to_str(x) for x in pile_to_seqs(s, rd, 200)])
print("== --trim windows (fa_batch_trim_windows) of %d E. coli-like piles" % len(piles))
b = eng.batch(piles)
b.trim_windows(8, 16)
t0 = time.perf_counter()
b.trim_windows(8, 16)
dt = time.perf_counter() - t0
n_reads = sum(len(p) - 1 for p in piles)
print("  %d reads in %.1f ms: %.0f reads/s, %.0f piles/s" % (n_reads, dt * 1e3, n_reads / dt, len(piles) / dt))
print("== batch ABI vs the legacy per-pile symbol on the same %d piles" % len(piles))
b.run(4, 8, 0.70)
t0 = time.perf_counter()
b.run(4, 8, 0.70).fetch(False)
dt_batch = time.perf_counter() - t0
want = [b.result(i) for i in range(len(piles))]
b.free()
from oracle.pyoracle import LegacyABI  # noqa: E402  (only its ctypes prototypes: drives the PRODUCT library)
legacy = LegacyABI(os.path.join(ROOT, "falcon_amd", "libfalcon_amd.so"))
legacy.generate_consensus(piles[0], 4, 8, 0.70)
t0 = time.perf_counter()
got = [legacy.generate_consensus(p, 4, 8, 0.70)[0] for p in piles[:64]]
dt_leg = time.perf_counter() - t0
assert got == want[:64]
print("  batch ABI: %d piles in %.1f ms = %.0f piles/s; legacy generate_consensus: 64 piles in %.2f s = %.1f piles/s "
      "(%.0fx slower: every call stages, runs and frees a batch of one behind a global lock)"
      % (len(piles), dt_batch * 1e3, len(piles) / dt_batch, dt_leg, 64 / dt_leg, (len(piles) / dt_batch) / (64 / dt_leg)))
eng.close()
