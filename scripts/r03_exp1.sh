#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
B="python bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-end-to-end"
for d in 0 1; do
  FALCON_AMD_A2_DEBUG=$d timeout 300 $B 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('debug=$d', r['kernel_ms']['k_align'], r['align']['pair_iterations'], r['align']['single_iterations'])"
done
