#!/bin/bash
# Round 5, the hand-scheduled row loop of k_align2 on the GPU box:
#   1. the shadow kernel (both renderings of the rows side by side, differences logged),
#   2. the alignment-facing part of the GPU suite on the product kernel,
#   3. a short bench line.
# usage: scripts/r05_a2.sh <tag> [full]
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for m in 1 2; do ( FALCON_AMD_A2_SHADOW=$m timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -s -k "align_golden_one_launch or synthetic_piles_vs_oracle or piles_golden_one_batch or ecoli_scale or every_hand_back" 2>&1 | grep -v "^$" | tail -40 ) > $O/shadow$m.txt; cat $O/shadow$m.txt; done
if [ "$2" = "full" ]; then
  ( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest_gpu.txt
else
  ( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_campaign.py -x -q 2>&1 | tail -25 ) > $O/pytest_gpu.txt
fi
cat $O/pytest_gpu.txt
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --steps 8 --warmup 2 > $O/bench.json.txt 2> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json.txt").read().strip().splitlines()[-1])
    print("value %.1f M" % (d["value"] / 1e6), "ms_per_step", d["ms_per_step"], "kernel_ms", d.get("kernel_ms"))
    print("align", d.get("align"))
    print("parity", d.get("parity_checked_piles"), d.get("parity_mismatches"))
except Exception as e:
    print("bench line unreadable:", e)
PY
