#!/bin/bash
# instruction counts of k_align2 only (one pass).  usage: scripts/r03_pmc_quick.sh <tag>
TAG=${1:-r03pmcq}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-include-regex "k_align" --output-format csv -d $O/p1 -o p1 -- $B > $O/p1.log 2>&1; echo "rc=$?"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_SALU --kernel-include-regex "k_align" --output-format csv -d $O/p2 -o p2 -- $B > $O/p2.log 2>&1; echo "rc=$?"
python $R/scripts/pmc_table.py $O > $O/pmc_table.txt 2>&1
find $O -name "*.db" -size +5M -delete
grep -A30 "^k_align2" $O/pmc_table.txt | head -40
