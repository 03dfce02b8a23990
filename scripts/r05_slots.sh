#!/bin/bash
# k_align2 with fewer resident wavefronts (FALCON_AMD_SLOTS): its own time one batch at a time, and
# the pipelined step -- does the consensus stage of the batch before fit beside it?
# usage: scripts/r05_slots.sh <tag>
TAG=${1:-r05slots}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for n in 8192 7168 6144 5120 4096; do
  FALCON_AMD_SLOTS=$n timeout 200 python bench.py --no-cpu-baseline --no-end-to-end --no-pipeline --steps 6 --warmup 2 > $O/serial_$n.json.txt 2> $O/serial_$n.err
  FALCON_AMD_SLOTS=$n timeout 200 python bench.py --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 > $O/piped_$n.json.txt 2> $O/piped_$n.err
  python - <<PY
import json
for f in ("serial", "piped"):
    try:
        d = json.loads(open("$O/%s_$n.json.txt" % f).read().strip().splitlines()[-1])
        print("slots $n", f, "ms_per_step %.2f" % d["ms_per_step"], "k_align %.2f" % d["kernel_ms"]["k_align"], "value %.1f M" % (d["value"] / 1e6))
    except Exception as e:
        print("slots $n", f, "unreadable", e)
PY
done | tee $O/summary.txt
