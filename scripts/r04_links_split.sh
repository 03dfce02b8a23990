#!/bin/bash
# where k_links2's time goes: the bench's one-batch-at-a-time kernel times with parts of the kernel switched off
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r04l}; mkdir -p $O; cd $R
for d in 0 1 2 3 4 7; do
  FALCON_AMD_LINKS_DEBUG=$d timeout 300 python bench.py --no-pipeline --no-cpu-baseline --no-end-to-end --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LINKS_DEBUG=$d k_links %.2f ms  k_score %.2f  k_tags %.2f' % (d['kernel_ms']['k_links'], d['kernel_ms']['k_score'], d['kernel_ms']['k_tags']))" | tee -a $O/links_split.txt
done
