#!/bin/bash
# instruction counts of k_align2 (one pass each): as is, and without the trace-back.  usage: scripts/r03_pmc_quick.sh <tag>
TAG=${1:-r03pmcq}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
for d in 0 1; do
export FALCON_AMD_A2_DEBUG=$d
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-include-regex "k_align" --output-format csv -d $O/d$d/p1 -o p1 -- $B > $O/d$d.p1.log 2>&1; echo "rc=$?"
python $R/scripts/pmc_table.py $O/d$d > $O/pmc_table_d$d.txt 2>&1
grep -A12 "^k_align2" $O/pmc_table_d$d.txt | head -14
done
find $O -name "*.db" -size +5M -delete
