#!/usr/bin/env python3
"""Multi-stream worker on one GPU: J jobs of N/J piles each through
falcon_amd.mains.consensus_multi, with one and with two engines (contexts) per device
(FALCON_AMD_ENGINES_PER_DEVICE) -- the measurement DESIGN.md 7.5 asks for.

    python scripts/exp_multistream.py [N=3072] [J=4]
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
piles = bench.gen_piles([1000003 + i for i in range(n)], 32)
tmp = tempfile.mkdtemp(prefix="multistream_")
per = -(-n // jobs)
argv = []
size = 0
for j in range(jobs):
    src = os.path.join(tmp, "piles_%d.txt" % j)
    with open(src, "wb") as f:
        bench.write_la4falcon(piles[j * per:(j + 1) * per], f)
    size += os.path.getsize(src)
    argv += ["--job", src, os.path.join(tmp, "cns_%d.fasta" % j)]
cmd = [sys.executable, "-m", "falcon_amd.mains.consensus_multi", "--output-multi", "--min-idt", "0.70",
       "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"] + argv
for per_gpu in ("1", "2", "1", "2"):
    t0 = time.time()
    subprocess.run(cmd, check=True, cwd=ROOT, stderr=subprocess.DEVNULL,
                   env=dict(os.environ, FALCON_AMD_ENGINES_PER_DEVICE=per_gpu, FALCON_AMD_DEVICES="0"))
    dt = time.time() - t0
    print("%d jobs, %s engine(s) per GPU: %d piles, %.0f MB in %.2f s = %.0f piles/s"
          % (jobs, per_gpu, n, size / 1e6, dt, n / dt), flush=True)
