#!/bin/bash
# On the GPU box: one serial and one pipelined bench line per library under gpurun_variants/.
TAG=${1:-r05policy}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for so in gpurun_variants/*.so; do
  n=$(basename $so .so)
  FALCON_AMD_LIB=$R/$so timeout 200 python bench.py --no-cpu-baseline --no-end-to-end --no-pipeline --steps 5 --warmup 2 > $O/$n.json.txt 2> $O/$n.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/$n.json.txt").read().strip().splitlines()[-1])
    a = d["align"]
    print("$n", "step %.2f" % d["ms_per_step"], "k_align %.2f" % d["kernel_ms"]["k_align"], "pair %d single %d replace %d park %d parity %s/%s" % (a["pair_iterations"], a["single_iterations"], a["replacements_in_loop"], a["parkings"], d.get("parity_checked_piles"), d.get("parity_mismatches")))
except Exception as e:
    print("$n unreadable", e)
PY
done | tee $O/summary.txt
