#!/bin/bash
# A consensus-stage kernel changed: the GPU tests, a kernel trace of one batch at a time, the pipelined line.
# usage: scripts/r05_links.sh <tag>
TAG=${1:-r05l}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kts -o kts -- python $R/bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/kts.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kts/*/*.db $O/kts/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt | cut -c1-140
find $O -name "*.db" -size +5M -delete
cd $R
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end > $O/bench_ecoli.json.txt 2> /dev/null; cut -c1-200 $O/bench_ecoli.json.txt
