#!/bin/bash
# usage: scripts/r04_step_vs_batch_size.sh <tag> : step time against batch size (pipelined, resident inputs)
TAG=${1:-r04q}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for n in 473 946 1536 3072; do
  timeout 300 python bench.py --piles $n --no-cpu-baseline --no-end-to-end --steps 12 --warmup 3 > $O/bench_$n.json.txt 2> $O/bench_$n.err
  timeout 300 python bench.py --piles $n --no-pipeline --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 > $O/bench_${n}_serial.json.txt 2>> $O/bench_$n.err
done
python - <<EOF
import json
for n in (473, 946, 1536, 3072):
    for suf in ("", "_serial"):
        try:
            d = json.loads(open("$O/bench_%d%s.json.txt" % (n, suf)).read().strip().splitlines()[-1])
            print(n, suf or "pipelined", "ms_per_step %.2f" % d["ms_per_step"], "us_per_pile %.2f" % (1e3 * d["ms_per_step"] / n), "piles/s %.0f" % d["piles_per_sec"], "kernel_ms", d.get("kernel_ms"))
        except Exception as e:
            print(n, suf, "unreadable:", e)
EOF
