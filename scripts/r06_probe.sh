#!/bin/bash
# Round 6, first GPU call: the vector pipe's cost table, HEAD's bench line on this box, and whether the wait / LDS
# counter passes complete when they are restricted to ONE kernel.   usage: scripts/r06_probe.sh <tag>
TAG=${1:-r06a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 300 scripts/ubench/valu_cost.bin > $O/ubench_valu_cost.txt 2>&1; tail -70 $O/ubench_valu_cost.txt
timeout 400 python bench.py --no-cpu-baseline --no-end-to-end > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; cut -c1-300 $O/bench_ecoli.json.txt
cd /tmp && export TMPDIR=/tmp
export FALCON_AMD_DEVICE_PACK=1
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"
for k in k_align2; do
  for i in 1 2; do
    eval ctrs=\$P$i
    mkdir -p $O/pmc_$k
    timeout 240 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "^$k" --output-format csv -d $O/pmc_$k/p$i -o p$i -- $B > $O/pmc_$k/p$i.log 2>&1; echo "pmc $k pass $i rc=$?"
  done
done
find $O -name "*.db" -size +5M -delete
find $O -name "*.csv" -size +2M -delete
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True)):
    tot = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        tot[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])] += float(row["Counter_Value"])
    for k, v in sorted(tot.items()):
        print("%-28s %-26s %.6g" % (k[0][:28], k[1], v))
PY
