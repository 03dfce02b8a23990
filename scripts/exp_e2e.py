#!/usr/bin/env python3
"""End-to-end rate of the consensus worker on a long stream (text on stdin -> FASTA on stdout):
N distinct E. coli-like piles written R times (new seed ids), so that start-up (process, HIP,
first buffers) is amortised the way a real .las block amortises it.

    python scripts/exp_e2e.py 3072 3 [ENV=VALUE[,ENV=VALUE] ...]     # one run per setting, plus the default
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
SETTINGS = [{}] + [dict(kv.split("=", 1) for kv in arg.split(",")) for arg in sys.argv[3:]]  # (A=1,B=2: one setting)


def one(job):
    r, s = job
    seed, rd = make_pile(1000003 + s, S=20000, coverage=40.0)
    return pile_to_la4falcon("%09d" % (r * N + s), seed, rd, 100000 * s + 1)


with mp.get_context("fork").Pool(32) as pool:
    chunks = pool.map(one, [(0, s) for s in range(N)], chunksize=4)
path = os.path.join(tempfile.gettempdir(), "e2e_stream.txt")
with open(path, "w") as f:
    for r in range(R):
        for s, c in enumerate(chunks):
            # same pile, new seed id (first token of the pile's first line)
            f.write("%09d" % (r * N + s) + c[9:])
    f.write("- -\n")
size = os.path.getsize(path)
opts = ["--output-multi", "--min-idt", "0.70", "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"]
ref_sha = None
for env in SETTINGS:
    t = time.time()
    with open(path) as fin, open(path + ".fa", "w") as fout, open(path + ".err", "w") as ferr:
        subprocess.run([sys.executable, "-m", "falcon_amd.mains.consensus", "-v", "1"] + opts, stdin=fin,
                       stdout=fout, stderr=ferr, check=True, cwd=ROOT,
                       env=dict(os.environ, FALCON_AMD_T_LAUNCH=repr(t), **env))
    dt = time.time() - t
    sha = hashlib.sha1(open(path + ".fa", "rb").read()).hexdigest()[:12]
    ref_sha = ref_sha or sha
    runs = [float(ln.split("GPU stages + download ")[1].split()[0]) for ln in open(path + ".err") if "GPU stages + download" in ln]
    print("%-44s %d piles, %.0f MB in %.2f s: %.0f piles/s, %.0f MB/s; %d batches, GPU stages + download "
          "%.0f ms mean; FASTA %s %s" % (",".join("%s=%s" % kv for kv in env.items()) or "default", N * R, size / 1e6, dt, N * R / dt, size / 1e6 / dt,
                                        len(runs), 1e3 * sum(runs) / max(1, len(runs)), sha,
                                        "(same)" if sha == ref_sha else "(DIFFERENT)"), flush=True)
    for ln in open(path + ".err"):
        if "steady state" in ln or "FALCON_AMD_TIMING" in ln:
            print("    " + ln.strip()[:300], flush=True)
if os.environ.get("EXP_E2E_KEEP_LOG"):
    import shutil
    shutil.copy(path + ".err", os.environ["EXP_E2E_KEEP_LOG"])
