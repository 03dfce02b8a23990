#!/bin/bash
# Round evidence on the GPU box: default bench line, rocprofv3 kernel-trace stats of the
# same command, and the HBM traffic counters in separate --pmc passes (kernel-trace only).
# usage: scripts/collect_profiles.sh <tag>      -> gpurun_out/<tag>/
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > $O/bench.json.txt 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $O/kt.log 2>&1
# (FETCH_SIZE and TCC_EA0_RDREQ_sum hang rocprofv3 on this pool until the timeout: left out)
for c in WRITE_SIZE TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum; do
  timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o $c -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $O/$c.log 2>&1
  echo "$c rc=$?"
done
tail -c 600 $O/bench.json.txt
