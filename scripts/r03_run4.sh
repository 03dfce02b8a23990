#!/bin/bash
# GPU tests (without the 65 s dmel/arab bench self-parity unless FULL=1) + a bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-t4}
if [ -n "$FULL" ]; then SEL=""; else SEL='-k not bench_self_parity'; fi
( timeout 900 python -m pytest tests -m gpu -q ${SEL:+"$SEL"} 2>&1 | tail -25 ) > gpurun_out/r03_${TAG}_pytest.txt
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r03_${TAG}_bench.json 2> gpurun_out/r03_${TAG}_bench.err
tail -3 gpurun_out/r03_${TAG}_pytest.txt
python - "$TAG" <<'P'
import json,sys
try:
    r=json.loads(open("gpurun_out/r03_%s_bench.json"%sys.argv[1]).read().strip().split("\n")[-1])
    print(r["value"], r["ms_per_step"], r["kernel_ms"], r.get("align"), r["roofline"].get("alone"), r.get("host_plan_gap_ms"))
except Exception as e:
    print("failed", e)
P
tail -5 gpurun_out/r03_${TAG}_bench.err
