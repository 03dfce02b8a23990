#!/bin/bash
# usage: scripts/r04_bench_full.sh <tag> : GPU suite + the default bench line (CPU baseline and end-to-end included)
TAG=${1:-r04b}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
( time timeout 900 python bench.py > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err ) 2>&1 | tail -4
python - <<EOF
import json
d = json.loads(open("$O/bench_ecoli.json.txt").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("alone"))
print("kernel_ms", d["kernel_ms"])
print("e2e", {k: v for k, v in d.get("end_to_end", {}).items() if k != "what"})
cb = d.get("cpu_baseline", {})
print("cpu", cb.get("value"), cb.get("cores"), cb.get("kind"), cb.get("per_core_bases_per_sec"))
for r in cb.get("runs", []): print("   ", r)
print("parity", d.get("parity_checked_piles"), d.get("parity_mismatches"), d.get("parity_against"))
print("align", d.get("align"))
EOF
tail -5 $O/bench_ecoli.err
