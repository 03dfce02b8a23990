#!/bin/bash
# round 3, first GPU call: parity of the new alignment kernel (k_align2) through the whole
# suite and the frozen campaign, then the bench with it and with the round-2 kernel on the same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r03_t1_pytest.txt
( timeout 300 python scripts/gpu_differential_campaign.py 2>&1 | tail -12 ) > gpurun_out/r03_t1_campaign.txt
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r03_t1_bench_a2.json 2> gpurun_out/r03_t1_bench_a2.err
FALCON_AMD_ALIGN1=1 timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r03_t1_bench_a1.json 2> gpurun_out/r03_t1_bench_a1.err
timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --no-pipeline > gpurun_out/r03_t1_bench_a2_nopipe.json 2> gpurun_out/r03_t1_bench_a2_nopipe.err
tail -3 gpurun_out/r03_t1_pytest.txt; tail -3 gpurun_out/r03_t1_campaign.txt
python - <<'P'
import json
for n in ("a2","a1","a2_nopipe"):
    try:
        r=json.loads(open("gpurun_out/r03_t1_bench_%s.json"%n).read().strip().split("\n")[-1])
        print(n, r["value"], r["ms_per_step"], r["kernel_ms"], r.get("align"), r["roofline"].get("alone"))
    except Exception as e:
        print(n, "failed", e)
P
