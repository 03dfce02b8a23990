"""bench.py, the synthetic input: SURVEY.md 8d's three workloads, the pile generator (falcon_amd.synth) in worker
processes, and the LA4Falcon text of a list of piles.  Not part of the product (nothing under falcon_amd/ imports it)."""
from __future__ import annotations

import multiprocessing as mp


HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


MIN_COV, K, MIN_IDT, MAX_N_READ = 4, 8, 0.70, 200


# SURVEY.md 8d configs 2/3, 4, 5; the flags are the same cfg line in all three
# (examples/fc_run_ecoli.cfg:33, fc_run_dmel.cfg:34, fc_run_arab.cfg:34)
WORKLOADS = {
    "ecoli": dict(S=20000, coverage=40.0, het=0.0, piles=3072,
                  text="E. coli-like piles: ~20 kb seed x 40x coverage, e=0.13 "
                       "(BASELINE.json configs[1]; falcon_amd/synth.py, SURVEY.md 8d)"),
    "dmel": dict(S=30000, coverage=80.0, het=0.0, piles=1024,
                 text="D. melanogaster-like piles: ~30 kb seed x 80x coverage, e=0.13, the "
                      "200-read cap binding (BASELINE.json configs[3]; SURVEY.md 8d config 4)"),
    "arab": dict(S=25000, coverage=60.0, het=0.005, piles=1536,
                 text="Arabidopsis-like piles: ~25 kb seed x 60x coverage, e=0.13, two haplotypes "
                      "0.5 % apart (BASELINE.json configs[4]; SURVEY.md 8d config 5)"),
}


def _gen_pile(job):
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    seed, S, cov, het = job
    s, rd = make_pile(seed, S=S, coverage=cov, het=het)
    return [codes_to_str(x).encode("ascii") for x in pile_to_seqs(s, rd, MAX_N_READ)]


def gen_piles(seeds, procs, wl):
    jobs = [(s, wl["S"], wl["coverage"], wl["het"]) for s in seeds]
    if procs <= 1 or len(jobs) < 4:
        return [_gen_pile(j) for j in jobs]
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        return pool.map(_gen_pile, jobs, chunksize=4)


def write_la4falcon(piles, f, repeats=1):
    """The LA4Falcon text a pile [seed, seed copy + reads by length] (falcon_amd.synth
    pile_to_seqs) came from: the seed line, then the reads (the reader adds the seed's copy
    itself, consensus.py:183-190).  `repeats`: the piles again under new seed ids, for a
    stream as long as a .las block's."""
    for rep in range(repeats):
        for i, p in enumerate(piles):
            lines, seen_copy = [b"%09d %s" % (rep * len(piles) + i, p[0])], False
            for j, r in enumerate(p[1:]):
                if not seen_copy and r == p[0]:
                    seen_copy = True
                    continue
                lines.append(b"%09d %s" % (1000000 + 1000 * i + j, r))
            f.write(b"\n".join(lines) + b"\n+ +\n")
    f.write(b"- -\n")
