R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc7
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc7/p1 -o p1 -- python $R/bench.py --steps 1 --warmup 0 --piles 768 --no-cpu-baseline > $R/gpurun_out/pmc7/p1.log 2>&1
echo rc=$?
