# PMC passes used for the k_links analysis in DESIGN.md (768 piles): wave/wait/issue cycles,
# L2 and L1 hit counters, one --pmc set per pass, kernel-trace only.
# NOTE: on this pool the passes with SQ_INST_CYCLES_SALU + SQ_ACTIVE_INST_VMEM (removed
# here), FETCH_SIZE and TCC_EA0_RDREQ_sum hang rocprofv3 until the timeout.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/p$i -o p$i -- python $R/bench.py --steps 1 --warmup 0 --piles 768 --no-cpu-baseline --no-end-to-end > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
