#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r03_t2_pytest.txt
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r03_t2_bench_a2.json 2> gpurun_out/r03_t2_bench_a2.err
tail -3 gpurun_out/r03_t2_pytest.txt
python - <<'P'
import json
for n in ("a2",):
    try:
        r=json.loads(open("gpurun_out/r03_t2_bench_%s.json"%n).read().strip().split("\n")[-1])
        print(n, r["value"], r["ms_per_step"], r["kernel_ms"], r.get("align"), r["roofline"].get("alone"))
    except Exception as e:
        print(n, "failed", e)
P
