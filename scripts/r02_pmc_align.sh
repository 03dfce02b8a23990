#!/bin/bash
# PMC passes of k_align alone at the bench batch size (kernel-trace only, one counter set per pass)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-pa}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "k_align" --output-format csv -d $O/p$i -o p$i -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-pipeline > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python $R/scripts/pmc_table.py $O | grep -v "^   TC"
