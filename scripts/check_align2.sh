#!/bin/bash
# k_align2 on the GPU box: the shadow kernel (both renderings of the rows side by side), the
# alignment-facing part of the GPU suite (or all of it), a short bench line.   usage: scripts/check_align2.sh <tag> [full]
TAG=${1:-r06b}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_campaign.py -x -q -k "hand_scheduled" 2>&1 | tail -60 ) > $O/shadow.txt; tail -40 $O/shadow.txt
if [ "$2" = "full" ]; then
  ( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/pytest_gpu.txt
else
  ( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_campaign.py -x -q 2>&1 | tail -25 ) > $O/pytest_gpu.txt
fi
cat $O/pytest_gpu.txt
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --steps 8 --warmup 2 > $O/bench.json.txt 2> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json.txt").read().strip().splitlines()[-1])
    print("value %.1f M" % (d["value"] / 1e6), "ms_per_step", d["ms_per_step"], "alone", d.get("kernel_ms_alone"))
    print("align", d.get("align"))
except Exception as e:
    print("bench line unreadable:", e)
PY
