#!/bin/bash
# The kernels' own times (three unpipelined steps) of the tree's library and of builds under gpurun_variants/, then the
# pipelined line of the tree's.   usage: scripts/ab_times.sh <tag> ["<variants>"]
TAG=${1:-r06q}; VARS=${2:-head}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for v in $VARS; do
  if [ "$v" = "head" ]; then unset FALCON_AMD_LIB; else export FALCON_AMD_LIB=$R/gpurun_variants/$v/libfalcon_amd.so; fi
  timeout 300 python bench.py --no-pipeline --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/serial_$v.json.txt 2> $O/serial_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/serial_$v.json.txt").read().strip().splitlines()[-1])
    print("%-12s" % "$v", " ".join("%s %.3f" % (k[2:], x) for k, x in d["kernel_ms"].items()), " sum %.2f" % sum(d["kernel_ms"].values()), " parity", d.get("parity_checked_piles"), d.get("parity_mismatches"))
except Exception as e:
    print("$v: bench line unreadable:", e, open("$O/serial_$v.err").read()[-500:])
PY
done
unset FALCON_AMD_LIB
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end > $O/bench.json.txt 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json.txt").read().strip().splitlines()[-1])
print("pipelined: value %.1f M  ms_per_step %.3f  parity %s / %s" % (d["value"] / 1e6, d["ms_per_step"], d.get("parity_checked_piles"), d.get("parity_mismatches")))
PY
