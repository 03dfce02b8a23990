#!/usr/bin/env python3
"""End-to-end rate of the consensus worker: LA4Falcon-style text on stdin -> FASTA on stdout
(native reader + staging overlapped with the GPU stages), against the python reader."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from falcon_amd.synth import make_pile, pile_to_la4falcon
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
def one(s):
    seed, rd = make_pile(1000003 + s, S=20000, coverage=40.0)
    return pile_to_la4falcon("%09d" % s, seed, rd, 100000 * s + 1)
import multiprocessing as mp
with mp.get_context("fork").Pool(32) as pool:
    chunks = pool.map(one, range(N), chunksize=4)
path = os.path.join(tempfile.gettempdir(), "cli_stream.txt")
with open(path, "w") as f:
    f.write("".join(chunks) + "- -\n")
size = os.path.getsize(path)
opts = ["--output-multi", "--min-idt", "0.70", "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"]
for label, env in (("native reader", {}), ("python reader", {"FALCON_AMD_PY_READER": "1"})):
    t = time.time()
    with open(path) as fin, open(path + ".fa", "w") as fout:
        subprocess.run([sys.executable, "-m", "falcon_amd.mains.consensus"] + opts, stdin=fin, stdout=fout,
                       check=True, cwd=ROOT, env=dict(os.environ, **env))
    dt = time.time() - t
    print("%s: %d piles, %.0f MB in %.2f s (incl. process start, torch-free): %.0f piles/s, %.0f MB/s, fasta %d bytes"
          % (label, N, size / 1e6, dt, N / dt, size / 1e6 / dt, os.path.getsize(path + ".fa")), flush=True)
