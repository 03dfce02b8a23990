#!/bin/bash
tag=${1:-bt}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 200 python scripts/gpu_differential_campaign.py piles 0 120 > $O/campaign.txt 2>&1; tail -1 $O/campaign.txt | cut -c1-200
bash scripts/r02_ab.sh $tag FALCON_AMD_BACKTRACE_WAVES=1
for s in "" FALCON_AMD_BACKTRACE_WAVES=1; do
env $s timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 --no-pipeline > $O/s.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/s.json').readline()); print('serial %-32s'%'$s', d['value'], d['ms_per_step'], d['kernel_ms'])"
done
