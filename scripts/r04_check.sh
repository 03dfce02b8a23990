#!/bin/bash
# Round-4 quick check on the GPU box: the GPU suite (or a -k subset), a short bench, per-kernel times.
# usage: scripts/r04_check.sh <tag> [pytest -k expression | "all" | "none"]
TAG=${1:-r04c}; KEXPR=${2:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ "$KEXPR" = "all" ]; then ( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt
elif [ "$KEXPR" != "none" ]; then ( timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" 2>&1 | tail -15 ) > $O/pytest_gpu.txt; fi
cat $O/pytest_gpu.txt 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; cut -c1-400 $O/bench_ecoli.json.txt; tail -3 $O/bench_ecoli.err
python - <<EOF
import json
try:
    d = json.loads(open("$O/bench_ecoli.json.txt").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "kernel_ms", d.get("kernel_ms"), "stage_ms", d.get("stage_ms"))
except Exception as e:
    print("bench line unreadable:", e)
EOF
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kts -o kts -- python $R/bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/kts.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kts/*/*.db $O/kts/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt 2>&1; head -16 $O/kernel_stats.txt | cut -c1-150
find $O -name "*.db" -size +5M -delete
