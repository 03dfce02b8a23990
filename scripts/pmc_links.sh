R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/p$i -o p$i -- python $R/bench.py --steps 1 --warmup 0 --piles 768 --no-cpu-baseline > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
