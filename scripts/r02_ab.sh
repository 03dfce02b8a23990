#!/bin/bash
# A/B of environment switches on the pipelined bench:  scripts/r02_ab.sh <tag> "ENV=1" "ENV2=1 ENV3=1" ...
tag=${1:-ab}; shift; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
i=0
for setting in "" "$@"; do
  i=$((i+1))
  env $setting timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --steps 8 --warmup 2 > $O/b$i.json 2> $O/b$i.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b$i.json").readline())
    print("%-40s" % ("$setting" or "default"), d["value"], d["ms_per_step"], d["kernel_ms"])
except Exception as e:
    print("$setting", "failed", e)
PY
done
