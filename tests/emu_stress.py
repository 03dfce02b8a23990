"""TEST INFRASTRUCTURE: stress run of the k_align2 source on the lane emulator: band-150 pairs
of the frozen campaign (oracle/campaign_cases.py), queued in random orders (= random
pairings and tape histories inside a wavefront) over an arena filled with random words,
each answer compared with the CPU oracle.

    python tests/emu_stress.py <first seed> <last seed> [trials] [ring]
    python tests/emu_stress.py synth <seed> [trials] [ring]     # reads like the bench's, 0-20 % apart
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

KEYS = ["dist", "aln_q_e", "aln_t_e", "aln_str_size", "q_aln_str", "t_aln_str", "cells"]


def main():
    from emu_driver import align_pairs
    from oracle.campaign_cases import function_cases
    from oracle.pyoracle import Port, build
    build()
    port = Port()
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    ring = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
    pairs, ids, windows = [], [], []
    if sys.argv[1] == "synth":
        from falcon_amd.synth import codes_to_str, noisy
        lo = hi = int(sys.argv[2])
        g = np.random.default_rng(lo)
        for n in range(48):
            L = int(g.choice([300, 900, 2500, 5000, 9000])) + int(g.integers(0, 400))
            e = float(g.choice([0.0, 0.03, 0.07, 0.10, 0.13, 0.16, 0.2]))
            base = g.integers(0, 4, L, dtype=np.uint8)
            q = codes_to_str(noisy(base, g, e))
            t = codes_to_str(noisy(base, g, e * float(g.choice([0.0, 0.5, 1.0]))))
            if g.random() < 0.3:  # tails that do not belong together
                q += codes_to_str(g.integers(0, 4, int(g.integers(1, 300)), dtype=np.uint8))
            if g.random() < 0.2:
                t = codes_to_str(g.integers(0, 4, int(g.integers(1, 200)), dtype=np.uint8)) + t
            pairs.append((q, t))
            ids.append(("synth", n))
            # (windows that begin and end anywhere: base offsets that are no multiples of 16)
            if g.random() < 0.5 and len(q) > 200 and len(t) > 200:
                a, b = int(g.integers(0, 40)), int(g.integers(0, 40))
                windows.append((a, len(q) - int(g.integers(0, 40)), b, len(t) - int(g.integers(0, 40))))
            else:
                windows.append((0, len(q), 0, len(t)))
    else:
        lo, hi = int(sys.argv[1]), int(sys.argv[2])
        for s in range(lo, hi + 1):
            for t, (q, tt, band) in enumerate(function_cases(s)):
                if band == 150:
                    pairs.append((q, tt))
                    ids.append((s, t))
    if not windows:
        windows = [(0, len(q), 0, len(t)) for q, t in pairs]
    want = [port.align(q[w[0]:w[1]], t[w[2]:w[3]], 150, 1) for (q, t), w in zip(pairs, windows)]
    print("%d pairs" % len(pairs), flush=True)
    rng = np.random.default_rng(lo * 1000 + hi)
    n_bad = 0
    for trial in range(trials):
        order = rng.permutation(2 * len(pairs)).astype(np.int32)
        os.environ["EMU_FILL"] = str(int(rng.integers(1, 1 << 30)))
        try:
            res, st = align_pairs(pairs, order=order, ring=ring, windows=windows)
        except Exception as exc:  # (a script that does not even expand)
            np.save("/tmp/emu_stress_order_%d_%d.npy" % (lo, trial), order)
            print("trial", trial, "broke:", repr(exc), "fill", os.environ["EMU_FILL"],
                  "order saved to /tmp/emu_stress_order_%d_%d.npy" % (lo, trial), flush=True)
            n_bad += 1
            continue
        back = 0
        for i, (r, o) in enumerate(zip(res, want)):
            if r["err"] == 2:
                back += 1
                continue
            if r["err"] != 0:
                print("trial", trial, ids[i], "err", r["err"])
                n_bad += 1
                continue
            if not r["aligned"]:
                if o["aln_str_size"] != 0 or r["cells"] != o["cells"]:
                    print("trial", trial, ids[i], "not aligned, oracle size", o["aln_str_size"])
                    n_bad += 1
                continue
            diff = [k for k in KEYS if r[k] != o[k]]
            if diff:
                pos = int(np.where(order == 2 * i + 1)[0][0])
                print("trial", trial, ids[i], "differs in", diff, "size", r["aln_str_size"], "want",
                      o["aln_str_size"], "queue position", pos, "fill", os.environ["EMU_FILL"])
                n_bad += 1
                np.save("/tmp/emu_stress_order_%d_%d.npy" % (lo, trial), order)
        print("trial %d: handed back %d, stats %s, bad so far %d" % (trial, back, list(map(int, st)), n_bad), flush=True)
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
