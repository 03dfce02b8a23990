"""DEVELOPMENT TOOL: the per-row band hulls of every alignment of bench-like piles, for
scripts/sim/a2_policy_sim.cpp (how k_align2's lane placement policies behave, counted on the host
before the kernel is touched).

    python scripts/sim/dump_bands.py --workload ecoli --piles 64 --out /tmp/bands_ecoli.bin

File: int32 n_aln, then per alignment (in the kernel's queue order: longest reads first,
engine.hip:680-688): int32 n_rows (rows that left a hull), int32 ok (1: ran to an end of a sequence),
int32 q_len, int32 t_len, int64 cells, then n_rows x {int32 lo, int32 hi} (extreme diagonals that
passed the band filter after the row)."""
import argparse
import ctypes
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from benchlib.workloads import WORKLOADS, gen_piles  # noqa: E402
from oracle.pyoracle import Port  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ecoli")
    ap.add_argument("--piles", type=int, default=64)
    ap.add_argument("--first-seed", type=int, default=1000)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    so = os.path.join(HERE, "libband_trace.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "band_trace.c")], check=True)
    lib = ctypes.CDLL(so)
    lib.band_trace.restype = ctypes.c_int
    lib.band_trace.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    wl = WORKLOADS[a.workload]
    piles = gen_piles(range(a.first_seed, a.first_seed + a.piles), 8, wl)
    port = Port()
    work = []  # (read length, q window, t window)
    for p in piles:
        seed = p[0]
        for r in p[1:]:
            h = port.find_hits(seed.decode(), r.decode())
            s1, e1, s2, e2, _score = port.best_range(h[0], h[1])
            # falcon.c:613-615
            if e1 - s1 < 100 or e2 - s2 < 100 or abs((e1 - s1) - (e2 - s2)) > int(0.5 * 0.10 * (e1 - s1 + e2 - s2)):
                continue
            work.append((len(r), r[s1:e1], seed[s2:e2]))
    work.sort(key=lambda w: -w[0])
    cap = 40000
    lo = np.zeros(cap, dtype=np.int32)
    hi = np.zeros(cap, dtype=np.int32)
    cells = ctypes.c_long(0)
    tot_rows = tot_cells = 0
    with open(a.out, "wb") as f:
        f.write(struct.pack("<i", len(work)))
        for _, q, t in work:
            n = lib.band_trace(q, len(q), t, len(t), 150, lo.ctypes.data, hi.ctypes.data, cap, ctypes.byref(cells))
            ok = 1 if n >= 0 else 0
            if n < 0:
                n = -n - 1
            f.write(struct.pack("<iiiiq", n, ok, len(q), len(t), cells.value))
            f.write(np.stack([lo[:n], hi[:n]], axis=1).astype("<i4").tobytes())
            tot_rows += n
            tot_cells += cells.value
    print("%d alignments, %d rows, %d cells (%.1f cells per row) -> %s" %
          (len(work), tot_rows, tot_cells, tot_cells / max(tot_rows, 1), a.out))


if __name__ == "__main__":
    main()
