#!/bin/bash
# The gate (engine.hip, fa_ctx::d_gate): batch n + 1's k_align2 held back until batch n's k_tags / k_links2 are
# through, then launched on fewer wavefront slots so that batch n's k_score2 / k_backtrace run beside it.
# usage: scripts/r05_gate.sh <tag>
TAG=${1:-r05gate}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for g in 0 2 4 6 8; do
  FALCON_AMD_GATE=$g timeout 200 python bench.py --no-cpu-baseline --no-end-to-end --steps 12 --warmup 3 > $O/gate_$g.json.txt 2> $O/gate_$g.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/gate_$g.json.txt").read().strip().splitlines()[-1])
    print("gate $g", "ms_per_step %.2f" % d["ms_per_step"], "value %.1f M" % (d["value"] / 1e6), "k_align %.2f" % d["kernel_ms"]["k_align"], "alone", d["roofline"]["alone"]["avg_launch_ms"], "slots", d["align"]["slots"])
except Exception as e:
    print("gate $g unreadable", e)
PY
done | tee $O/summary.txt
FALCON_AMD_GATE=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pipelined or bench_scale or order_preserving or piles_golden_one_batch" 2>&1 | tail -3
