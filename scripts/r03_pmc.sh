#!/bin/bash
# instruction / wait / cache counters of the alignment kernel: k_align2 (default) and, with
# FALCON_AMD_ALIGN1=1, the round-2 kernel on the same box.  usage: scripts/r03_pmc.sh <tag>
TAG=${1:-r03pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
for which in a2 a1; do
  if [ $which = a1 ]; then export FALCON_AMD_ALIGN1=1; else unset FALCON_AMD_ALIGN1; fi
  mkdir -p $O/$which
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
              "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
              "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
              "WRITE_SIZE" "FETCH_SIZE"; do
    i=$((i+1))
    timeout 150 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "k_align|k_pack" --output-format csv -d $O/$which/p$i -o p$i -- $B > $O/$which/p$i.log 2>&1; echo "$which pass $i rc=$?"
  done
  python $R/scripts/pmc_table.py $O/$which > $O/pmc_table_$which.txt 2>&1
  find $O/$which -name "*.db" -size +5M -delete
done
cat $O/pmc_table_a2.txt | head -60
cat $O/pmc_table_a1.txt | head -60
