#!/bin/bash
# collect PMC counters for the bench kernels: separate passes, kernel-trace only (no sys-trace)
# usage: scripts/pmc_run.sh <tag> <piles> <pass-set: sq|mem|all>
TAG=${1:-pmc}; PILES=${2:-768}; SET=${3:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
SQ2="SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"
SQ3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT"
declare -a PASSES
case $SET in
  sq) PASSES=("$SQ1" "$SQ2" "$SQ3");;
  # (a FETCH_SIZE pass hangs rocprofv3 on this pool until the timeout: only the write side)
  mem) PASSES=("WRITE_SIZE");;
  *) PASSES=("$SQ1" "$SQ2" "$SQ3" "WRITE_SIZE");;
esac
i=0
for ctrs in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/$TAG/p$i -o p$i -- python $R/bench.py --steps 1 --warmup 0 --piles $PILES --no-cpu-baseline --no-end-to-end > $R/gpurun_out/$TAG/p$i.log 2>&1
  grep -i "error\|invalid\|unknown" $R/gpurun_out/$TAG/p$i.log | head -3
done
