#!/bin/bash
# usage: scripts/r04_quick.sh <tag> : a handful of parity tests and the kernels' own times (three workloads, one batch at a time)
TAG=${1:-r04v}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_one_batch or synthetic or fallback or deep or long_insertion" 2>&1 | tail -3 ) > $O/pytest.txt; cat $O/pytest.txt
for w in ecoli dmel arab; do
  timeout 120 python bench.py --workload $w --no-pipeline --no-cpu-baseline --no-end-to-end --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'ms_per_step', d['ms_per_step'], d['kernel_ms'])"
done
