#!/bin/bash
# Round-4 check on the GPU box: the GPU suite, the default bench line (with the end-to-end leg, without the CPU
# baseline), per-kernel times of the three workloads one batch at a time.
# usage: scripts/r04_check.sh <tag>
TAG=${1:-r04k}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 420 python bench.py --no-cpu-baseline > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; tail -3 $O/bench_ecoli.err
for w in ecoli dmel arab; do
  timeout 300 python bench.py --workload $w --no-pipeline --no-cpu-baseline --no-end-to-end --steps 4 --warmup 1 > $O/bench_${w}_serial.json.txt 2> $O/bench_${w}_serial.err
done
python - <<EOF
import json
for f in ("bench_ecoli", "bench_ecoli_serial", "bench_dmel_serial", "bench_arab_serial"):
    try:
        d = json.loads(open("$O/%s.json.txt" % f).read().strip().splitlines()[-1])
        print(f, "value %.1f M" % (d["value"] / 1e6), "ms_per_step", d["ms_per_step"], "kernel_ms", d.get("kernel_ms"))
        e = d.get("end_to_end")
        if e: print("   e2e", e.get("piles_per_sec"), e.get("runs_wall_s"), e.get("worker_steady_state_piles_per_sec"), e.get("text_MB_per_sec"))
    except Exception as e:
        print(f, "unreadable:", e)
EOF
