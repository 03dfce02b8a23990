"""Level / link statistics of a bench pile's MSA graph (what k_score walks), from the CPU oracle.
usage: python scripts/msa_level_stats.py [workload] [seed]   (test infrastructure: uses oracle/)"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import WORKLOADS, _gen_pile, MIN_IDT
from oracle.pyoracle import Port

wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "ecoli"]
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pile = [s.decode() for s in _gen_pile((seed, wl["S"], wl["coverage"], wl["het"]))]
P = Port()
t = pile[0]
T = len(t)
levels = collections.defaultdict(lambda: collections.defaultdict(int))  # (t,d) -> {(base,pt,pd,pb): count}
cov = np.zeros(T + 1, int)
nacc = 0
for q in pile[1:]:
    hq, ht = P.find_hits(t, q)
    s1, e1, s2, e2, sc = P.best_range(hq, ht)
    if e1 - s1 < 500 or e2 - s2 < 500: continue  # (approximate filter; statistics only)
    a = P.align(q[s1:e1], t[s2:e2])
    if a["aln_str_size"] < 500 or a["dist"] / a["aln_str_size"] >= 1 - MIN_IDT: continue
    nacc += 1
    i, j, jj = s1 - 1, s2 - 1, 0
    pt, pd, pb = -1, 0, '.'
    for qc, tc in zip(a["q_aln_str"], a["t_aln_str"]):
        if qc != '-': i += 1; jj += 1
        if tc != '-': j += 1; jj = 0
        levels[(j, jj)][(qc, pt, pd, pb)] += 1
        pt, pd, pb = j, jj, qc
    cov[s2] += 1; cov[j + 1] -= 1
nl = collections.Counter(len(v) for v in levels.values())
per_t = collections.Counter()
for (tp, d) in levels: per_t[tp] = max(per_t[tp], d + 1)
nlev = collections.Counter(per_t.values())
tot = len(levels)
print("accepted", nacc, "T", T, "levels", tot, "links", sum(len(v) for v in levels.values()))
print("levels per position:", sorted(nlev.items()))
print("links per level:", sorted(nl.items()))
print("levels with > 16 links: %.4f %%" % (100.0 * sum(c for n, c in nl.items() if n > 16) / tot))
nodes = collections.Counter(len(set(k[0] for k in v)) for v in levels.values())
print("nodes per level:", sorted(nodes.items()))
# boundary candidates: positions whose previous position has a single level and a single delta-0 node
one = sum(1 for tp in per_t if per_t.get(tp - 1, 0) == 1)
one1 = sum(1 for tp in per_t if per_t.get(tp - 1, 0) == 1 and len(set(k[0] for k in levels[(tp - 1, 0)])) == 1)
print("positions after a 1-level position: %d, after a 1-level 1-node position: %d" % (one, one1))
