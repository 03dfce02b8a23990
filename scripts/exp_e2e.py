#!/usr/bin/env python3
"""Experiment: stalls of the consensus worker when staging overlaps the GPU stages."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
piles = bench.gen_piles([1000003 + i for i in range(n)], 32)
src = "/tmp/e2e_piles.txt"
with open(src, "wb") as f:
    bench.write_la4falcon(piles, f)
cmd = [sys.executable, "-m", "falcon_amd.mains.consensus", "--output-multi", "--min-idt", "0.70",
       "--min-cov", "4", "--max-n-read", "200", "--n-core", "1", "--verbose-level", "1"]
for label, env in (("two streams", {}), ("one stream", {"FALCON_AMD_ONE_STREAM": "1"}),
                   ("two streams", {}), ("one stream", {"FALCON_AMD_ONE_STREAM": "1"}),
                   ("two streams", {}), ("one stream", {"FALCON_AMD_ONE_STREAM": "1"})):
    t0 = time.time()
    with open(src) as fin, open("/tmp/e2e.fa", "w") as fout:
        r = subprocess.run(cmd, stdin=fin, stdout=fout, stderr=subprocess.PIPE, text=True, check=True,
                           cwd=ROOT, env=dict(os.environ, **env))
    wall = time.time() - t0
    st = [float(x) for x in re.findall(r"staged in ([0-9.]+) s", r.stderr)]
    gp = [float(x) for x in re.findall(r"fetch ([0-9.]+) s", r.stderr)]
    print("%-12s wall %.2f s  staged %s  gpu %s" % (label, wall, st, gp), flush=True)
