#!/bin/bash
# Timelines (rocprofv3 kernel trace, scripts/timeline.py) of three pipelined steps with the gate off and on.
TAG=${1:-r05tl}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in 0 4; do
  FALCON_AMD_GATE=$g timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/g$g -o kt -- python $R/bench.py --no-cpu-baseline --no-end-to-end --steps 8 --warmup 2 > $O/g$g.log 2>&1
  f=$(find $O/g$g -name "*kernel_trace.csv" | head -1)
  python $R/scripts/timeline.py $f - 5 3 > $O/timeline_g$g.txt 2>&1
done
find $O -name "*.csv" -size +3M -delete
