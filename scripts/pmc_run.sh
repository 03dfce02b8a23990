#!/bin/bash
# collect PMC counters for the bench kernels: separate passes, kernel-trace only (no sys-trace)
# usage: scripts/pmc_run.sh <tag> <piles>
TAG=${1:-pmc}; PILES=${2:-768}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
            "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/$TAG/p$i -o p$i -- python $R/bench.py --steps 1 --warmup 0 --piles $PILES --no-cpu-baseline > $R/gpurun_out/$TAG/p$i.log 2>&1
done
find $R/gpurun_out/$TAG -name "*.csv" | head -20
