#!/usr/bin/env python3
"""Batch-size policy of the consensus worker (falcon_amd/mains/consensus.py _run_native):
end-to-end rate on N piles of text for several batch sizes (FALCON_AMD_BATCH_BASES).
(profiles/r01_v8_batch_policy.txt also has the doubling ramp that was tried and dropped.)"""
import os, sys, tempfile, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
piles = bench.gen_piles([1000003 + i for i in range(n)], 32)
src = os.path.join(tempfile.gettempdir(), "policy_piles.txt")
with open(src, "wb") as f:
    bench.write_la4falcon(piles, f)
size = os.path.getsize(src)
cmd = [sys.executable, "-m", "falcon_amd.mains.consensus", "--output-multi", "--min-idt", "0.70",
       "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"]
for label, env in (("0.4 G (default)", {}), ("0.8 G", {"FALCON_AMD_BATCH_BASES": "800000000"}),
                   ("1.3 G", {"FALCON_AMD_BATCH_BASES": "1300000000"}), ("0.2 G", {"FALCON_AMD_BATCH_BASES": "200000000"})):
    t0 = time.time()
    with open(src) as fin, open(src + ".fa", "w") as fout:
        subprocess.run(cmd, stdin=fin, stdout=fout, stderr=subprocess.DEVNULL, check=True, cwd=ROOT,
                       env=dict(os.environ, **env), timeout=60)
    dt = time.time() - t0
    print("%-18s %d piles, %.0f MB: %.2f s, %.0f piles/s" % (label, n, size / 1e6, dt, n / dt), flush=True)
