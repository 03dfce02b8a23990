#!/usr/bin/env python3
"""Experiment behind DESIGN.md 7.1: one context running all piles against two contexts (two
host threads, two streams) running half of them each on the same GPU; FALCON_AMD_SLOTS caps
the resident k_align wavefronts."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
N = 3072
piles = bench.gen_piles([1000003 + i for i in range(N)], 32)
import torch
torch.cuda.set_device(0)
from falcon_amd.engine import Engine
def timed(parts, reps=4, label=""):
    engs = [Engine(0) for _ in parts]
    bats = [e.batch(p) for e, p in zip(engs, parts)]
    def work(b):
        for _ in range(reps):
            b.run(4, 8, 0.70)
    for b in bats: b.run(4, 8, 0.70)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(b,)) for b in bats]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%s %d stream(s): %.1f ms per %d piles -> %.0f piles/s" % (label, len(parts), dt * 1e3, N, N / dt), flush=True)
    for b in bats:
        st = b.stats()
        print("    index %.1f chain %.1f align %.1f tags %.1f links %.1f score %.1f bt %.1f" % (st.ms_index, st.ms_chain, st.ms_align, st.ms_tags, st.ms_links, st.ms_score, st.ms_backtrace), flush=True)
    for b in bats: b.free()
    for e in engs: e.close()
slots = os.environ.get("FALCON_AMD_SLOTS", "all")
timed([piles], label="slots=%s" % slots)
h = N // 2
timed([piles[:h], piles[h:]], label="slots=%s" % slots)
