#!/bin/bash
# kernel timeline of pipelined bench steps: every dispatch with its stream/queue, start and gaps
tag=${1:-tl}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/kt.log 2>&1
ls -R $O/kt | head -20
python $R/scripts/timeline.py $(find $O/kt -name "*kernel_trace.csv" | head -1) $(find $O/kt -name "*memory_copy_trace.csv" | head -1) > $O/timeline.txt 2>&1
tail -60 $O/timeline.txt
find $O -name "*.db" -size +20M -delete
