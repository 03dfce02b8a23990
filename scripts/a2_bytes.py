#!/usr/bin/env python3
"""DEVELOPMENT TOOL: what k_align2 reads and writes, by what is accessed, counted on the host.

The kernel's source runs on the lane emulator (tests/emu) over the alignments of bench-like piles in the kernel's
queue order; every wave-wide load / store of the source is counted with the bytes its lanes touch and the 32 / 64 /
128-byte units they fall into, by region: packed words, tape cells, tape records, escape list, edit scripts,
alignment records.  Scaled to a launch of `--scale-to` piles and set against the counters on file
(profiles/pmc_traffic.json).  What the emulator cannot see: the register spills of the device build (scratch), and
that the device's rows read the sequences out of LDS windows (the emulator's rows read them from memory: its
"packed words" line is NOT the device's).

    python scripts/a2_bytes.py --piles 6 > profiles/r06_k_align2_bytes_by_site.txt
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import emu_driver  # noqa: E402
from benchlib.workloads import WORKLOADS, gen_piles  # noqa: E402
from oracle.pyoracle import Port  # noqa: E402

WHAT = ["packed words (emulator's rows: see above)", "tape: cells (1 B per lane and iteration)",
        "tape: records (16 B per iteration)", "tape: escape list", "edit scripts (4 B per band row)",
        "alignment records"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ecoli")
    ap.add_argument("--piles", type=int, default=6)
    ap.add_argument("--first-seed", type=int, default=1000)
    ap.add_argument("--scale-to", type=int, default=3072)
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]
    piles = gen_piles(range(a.first_seed, a.first_seed + a.piles), 8, wl)
    port = Port()
    work = []
    for p in piles:
        seed = p[0].decode()
        for r in p[1:]:
            r = r.decode()
            h = port.find_hits(seed, r)
            if not h[0]:
                continue
            s1, e1, s2, e2, _score = port.best_range(h[0], h[1])
            if e1 - s1 < 100 or e2 - s2 < 100 or abs((e1 - s1) - (e2 - s2)) > int(0.5 * 0.10 * (e1 - s1 + e2 - s2)):
                continue  # falcon.c:613-619
            work.append((len(r), r, seed, (s1, e1, s2, e2)))
    work.sort(key=lambda w: -w[0])  # the kernel's queue: longest reads first (engine.hip)
    lib = emu_driver.lib()
    lib.emu_acct_on(1)
    t0 = time.time()
    res, stats = emu_driver.align_pairs([(w[1], w[2]) for w in work], windows=[w[3] for w in work], ring=8192)
    tab = np.zeros((6, 2, 5))
    lib.emu_acct_get(tab.ctypes.data_as(C.c_void_p))
    lib.emu_acct_on(0)
    k = a.scale_to / a.piles
    rows = sum(r["dist"] + 1 for r in res if r["aligned"])
    cells = sum(r["cells"] for r in res)
    its = int(stats[0]) + int(stats[1])
    print("k_align2 by what it accesses: the kernel's source on the lane emulator, %d %s-like piles (%d alignments, %.0f s),"
          % (a.piles, a.workload, len(work), time.time() - t0))
    print("scaled x %.0f to a launch of %d piles.  Emulated: %d iterations (%d with two tracks), %d band rows, %d cells."
          % (k, a.scale_to, its, int(stats[0]), rows, cells))
    print("Per launch: %.1f M iterations, %.1f M band rows, %.2f G cells   (the bench line's `align` / `work` counts: 336.5 M, 578.3 M, 15.75 G)"
          % (its * k / 1e6, rows * k / 1e6, cells * k / 1e9))
    print()
    print("%-46s %-6s %10s %10s %10s %10s %10s" % ("what", "", "instr (M)", "bytes GB", "32 B GB", "64 B GB", "128 B GB"))
    tot = np.zeros((2, 5))
    for w in range(6):
        for rw in (1, 0):
            t = tab[w][rw] * k
            if t[0] == 0:
                continue
            if w != 0:
                tot[rw] += t
            print("%-46s %-6s %10.1f %10.2f %10.2f %10.2f %10.2f" % (WHAT[w], "write" if rw else "read", t[0] / 1e6, t[1] / 1e9,
                                                                      t[2] * 32 / 1e9, t[3] * 64 / 1e9, t[4] * 128 / 1e9))
    for rw in (1, 0):
        print("%-46s %-6s %10.1f %10.2f %10.2f %10.2f %10.2f" % ("all but the packed words", "write" if rw else "read", tot[rw][0] / 1e6,
                                                                  tot[rw][1] / 1e9, tot[rw][2] * 32 / 1e9, tot[rw][3] * 64 / 1e9,
                                                                  tot[rw][4] * 128 / 1e9))
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        r = rec.get("k_align", rec)
        print()
        print("counters on file (profiles/pmc_traffic.json, taken on %s): written %.1f GB, read %.1f GB per launch"
              % (r.get("taken_on", "?"), r.get("write_bytes", 0) / 1e9, r.get("read_bytes", 0) / 1e9))
    except (OSError, ValueError, AttributeError):
        pass


if __name__ == "__main__":
    main()
