#!/bin/bash
# usage: scripts/r04_exp3.sh <tag> : k_align2 with fewer resident wavefronts (room for the consensus stage beside it),
# the per-kernel times one batch at a time, the worker's timeline on the 30720-pile stream
TAG=${1:-r04m}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "synthetic or golden_one_batch or fallback" 2>&1 | tail -4 ) > $O/pytest.txt; cat $O/pytest.txt
for s in default 7168 6144 5120 4096; do
  if [ $s = default ]; then unset FALCON_AMD_SLOTS; else export FALCON_AMD_SLOTS=$s; fi
  timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --steps 8 --warmup 2 > $O/bench_slots_$s.json.txt 2> $O/bench_slots_$s.err
done
unset FALCON_AMD_SLOTS
timeout 300 python bench.py --no-pipeline --no-cpu-baseline --no-end-to-end --steps 4 --warmup 1 > $O/bench_serial.json.txt 2> $O/bench_serial.err
python - <<EOF
import json
for f in ("slots_default", "slots_7168", "slots_6144", "slots_5120", "slots_4096", "serial"):
    try:
        d = json.loads(open("$O/bench_%s.json.txt" % f).read().strip().splitlines()[-1])
        print(f, "value %.1f M" % (d["value"] / 1e6), "ms_per_step", d["ms_per_step"], "alone", (d["roofline"].get("alone") or {}).get("avg_launch_ms"), "kernel_ms", d.get("kernel_ms"))
    except Exception as e:
        print(f, "unreadable:", e)
EOF
FALCON_AMD_TIMING=1 timeout 600 python scripts/exp_e2e.py 3072 10 FALCON_AMD_NOTHING=1 > $O/e2e.txt 2>&1; cat $O/e2e.txt | cut -c1-250
head -70 /tmp/e2e_stream.txt.err > $O/e2e_timeline_head.txt; tail -40 /tmp/e2e_stream.txt.err > $O/e2e_timeline_tail.txt
grep -c . /tmp/e2e_stream.txt.err
