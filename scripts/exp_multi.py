#!/usr/bin/env python3
"""The multi-stream worker on one GPU: J jobs (files of N/J E. coli-like piles each) through ONE
`falcon_amd.mains.consensus_multi` process against the same piles as one stream through
`falcon_amd.mains.consensus`: wall time, piles/s, and that every job's FASTA equals the
corresponding part of the single stream's.

    python scripts/exp_multi.py 3072 3 4        # N distinct piles, R repeats, J jobs
"""
import hashlib
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from falcon_amd.synth import make_pile, pile_to_la4falcon  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
J = int(sys.argv[3]) if len(sys.argv) > 3 else 4


def one(s):
    seed, rd = make_pile(1000003 + s, S=20000, coverage=40.0)
    return pile_to_la4falcon("%09d" % s, seed, rd, 100000 * s + 1)


with mp.get_context("fork").Pool(32) as pool:
    chunks = pool.map(one, range(N), chunksize=4)
tmp = tempfile.gettempdir()
total = N * R
per_job = -(-total // J)
paths, n = [], 0
files = [open(os.path.join(tmp, "multi_%d.txt" % j), "w") for j in range(J)]
whole = open(os.path.join(tmp, "multi_all.txt"), "w")
for r in range(R):
    for s, c in enumerate(chunks):
        text = "%09d" % (r * N + s) + c[9:]
        files[n // per_job].write(text)
        whole.write(text)
        n += 1
for f in files + [whole]:
    f.write("- -\n")
    f.close()
opts = ["--output-multi", "--min-idt", "0.70", "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"]
env = dict(os.environ, PYTHONPATH=ROOT)

t = time.time()
with open(whole.name) as fin, open(whole.name + ".fa", "w") as fout:
    subprocess.run([sys.executable, "-m", "falcon_amd.mains.consensus"] + opts, stdin=fin, stdout=fout,
                   check=True, cwd=ROOT, env=env)
t_single = time.time() - t
single = open(whole.name + ".fa", "rb").read()

cmd = [sys.executable, "-m", "falcon_amd.mains.consensus_multi"] + opts
for j in range(J):
    cmd += ["--job", files[j].name, files[j].name + ".fa"]
t = time.time()
subprocess.run(cmd, check=True, cwd=ROOT, env=env)
t_multi = time.time() - t
multi = b"".join(open(files[j].name + ".fa", "rb").read() for j in range(J))
print("one stream : %d piles in %.2f s = %.0f piles/s" % (total, t_single, total / t_single))
print("%d streams  : %d piles in %.2f s = %.0f piles/s (one process, one GPU); FASTA of the jobs, concatenated, %s "
      "the single stream's (%s)" % (J, total, t_multi, total / t_multi,
                                    "equals" if multi == single else "DIFFERS FROM",
                                    hashlib.sha1(single).hexdigest()[:12]), flush=True)
