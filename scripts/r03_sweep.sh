#!/bin/bash
# one short bench line per library variant under gpurun_variants/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03sweep
export TMPDIR=/tmp
for so in gpurun_variants/libfalcon_amd_*.so; do
    name=$(basename $so .so); name=${name#libfalcon_amd_}
    FALCON_AMD_LIB=$PWD/$so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r03sweep/$name.json 2> gpurun_out/r03sweep/$name.err
    python - "$name" <<'P'
import json,sys
try:
    r=json.loads(open("gpurun_out/r03sweep/%s.json"%sys.argv[1]).read().strip().split("\n")[-1])
    a=r["align"]
    print("%-14s %.1f M  step %.2f  align %.2f alone %.2f  pair %d single %d place %d park %d repl %d back %d" % (sys.argv[1], r["value"]/1e6, r["ms_per_step"], r["kernel_ms"]["k_align"], r["roofline"]["alone"]["avg_launch_ms"], a["pair_iterations"]/1e6, a["single_iterations"]/1e6, a["placements"]/1e3, a["parkings"]/1e3, a["replacements_in_loop"]/1e3, a["handed_back"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
P
done
