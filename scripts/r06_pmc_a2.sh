#!/bin/bash
# k_align2's instruction counters (one restricted pass each) and its time alone.   usage: scripts/r06_pmc_a2.sh <tag>
TAG=${1:-r06c}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FALCON_AMD_DEVICE_PACK=1
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS"
for i in 3 1; do
  eval ctrs=\$P$i
  timeout 240 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "^k_align2" --output-format csv -d $O/p$i -o p$i -- $B > $O/p$i.log 2>&1; echo "pass $i rc=$?"
done
find $O -name "*.db" -size +5M -delete
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/**/*counter_collection.csv", recursive=True)):
    tot = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        tot[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])] += float(row["Counter_Value"])
    for k, v in sorted(tot.items()):
        print("%-28s %-26s %.6g" % (k[0][:28], k[1], v))
PY
