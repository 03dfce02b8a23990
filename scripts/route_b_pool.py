#!/usr/bin/env python3
"""Route (b) of INTEGRATION.md the way `north_star` words it: the reference's own driver keeps
its process pool (falcon_kit/mains/consensus.py:264,274: `Pool(n_core)` + `imap` of one pile
per task, each task one `falcon.generate_consensus` call through ctypes) and only the shared
library behind `falcon_kit.falcon_kit` is swapped for libfalcon_amd.so.  Every worker process
then owns a HIP context on the same GPU and sends it batches of ONE pile.

This script restates that calling pattern (a pool of n_core forked workers, piles handed out
one at a time through imap, the ctypes marshalling of consensus.py:102-120) on synthetic
E. coli-like piles and reports piles/s and the VRAM the n_core contexts hold, for
n_core = 1, 6, 24 -- beside the batch engine's rate on the same piles, whose strings every
worker's answers must equal.

    python scripts/route_b_pool.py [n_piles] [n_core ...]
"""
import glob
import multiprocessing as mp
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def vram_used():
    tot = 0
    for f in glob.glob("/sys/class/drm/card*/device/mem_info_vram_used"):
        try:
            tot += int(open(f).read())
        except (OSError, ValueError):
            pass
    return tot


def make(seed):
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    s, rd = make_pile(seed, S=20000, coverage=40.0)
    return [codes_to_str(x).encode() for x in pile_to_seqs(s, rd, 200)]


def one_pile(job):
    """What a pool worker of the reference's driver does with a pile (consensus.py:102-120)."""
    from ctypes import c_char_p, string_at
    from falcon_amd import falcon_kit as fk   # = what `from falcon_kit import falcon` resolves to (dropin/)
    seqs, seed_id, (min_cov, K, min_idt) = job
    ptr_arr = (c_char_p * len(seqs))()
    ptr_arr[:] = seqs
    cd = fk.falcon.generate_consensus(ptr_arr, len(seqs), min_cov, K, min_idt)
    cns = string_at(cd[0].sequence)[:]
    fk.falcon.free_consensus_data(cd)
    return seed_id, cns


def main():
    n_piles = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    cores = [int(x) for x in sys.argv[2:]] or [1, 6, 24]
    ctx = mp.get_context("fork")
    with ctx.Pool(min(32, os.cpu_count() or 1)) as pool:
        piles = pool.map(make, range(9000, 9000 + n_piles), chunksize=4)
    cfg = (4, 8, 0.70)
    jobs = [(p, "%09d" % i, cfg) for i, p in enumerate(piles)]
    base = vram_used()
    results = {}
    for n_core in cores:
        peak = [0]
        stop = threading.Event()

        def watch():
            while not stop.is_set():
                peak[0] = max(peak[0], vram_used())
                time.sleep(0.02)
        th = threading.Thread(target=watch, daemon=True)
        th.start()
        t0 = time.perf_counter()
        with ctx.Pool(n_core) as pool:   # (forked before any HIP call of this process: each worker opens its own context)
            out = list(pool.imap(one_pile, jobs))
        wall = time.perf_counter() - t0
        stop.set()
        th.join()
        results[n_core] = dict(out)
        print("route (b), Pool(%2d): %d piles in %.2f s = %.1f piles/s; VRAM held by the workers at the peak: "
              "%.2f GB (%.2f GB per process)" % (n_core, len(jobs), wall, len(jobs) / wall,
                                                 (peak[0] - base) / 1e9, (peak[0] - base) / 1e9 / n_core), flush=True)
    # the batch engine on the same piles (one process, one batch), and the identity of every answer
    from falcon_amd.engine import Engine
    eng = Engine(0)
    t0 = time.perf_counter()
    batch = eng.consensus([[s.decode() for s in p] for p in piles], 4, 8, 0.70)
    wall = time.perf_counter() - t0
    eng.close()
    print("batch engine, same piles: %.2f s = %.0f piles/s (staging from python lists included)" % (wall, len(piles) / wall))
    for n_core, res in results.items():
        bad = [i for i in range(len(piles)) if res["%09d" % i].decode() != batch[i]]
        print("Pool(%d) answers identical to the batch engine's: %s" % (n_core, "yes" if not bad else "NO: %r" % bad[:8]))


if __name__ == "__main__":
    main()
