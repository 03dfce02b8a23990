#!/bin/bash
# quick GPU check of a kernel change: parity suite (-x), the differential campaign, a bench line.
# usage: scripts/r02_quick.sh <tag> [extra bench args]
TAG=${1:-q}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 200 python scripts/gpu_differential_campaign.py piles 0 72 > $O/campaign_piles.txt 2>&1; tail -1 $O/campaign_piles.txt | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end "$@" > $O/bench.json.txt 2> $O/bench.err; python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json.txt").readline())
    print(d["value"], d["ms_per_step"], d["kernel_ms"])
except Exception as e:
    print("bench failed", e); print(open("$O/bench.err").read()[-2000:])
PY
