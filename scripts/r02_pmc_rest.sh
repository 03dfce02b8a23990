#!/bin/bash
# the two counter passes the evidence run lost to a rocprofv3 hang, restricted to the path's kernels
TAG=${1:-r02h}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
K="k_align|k_links|k_score|k_tags|k_chain|k_backtrace|k_seed_index|k_pack"
timeout 110 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-include-regex "$K" --output-format csv -d $O/pmc/p1 -o p1 -- $B > $O/pmc/p1.log 2>&1; echo "pass 1 rc=$?"
timeout 110 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-include-regex "$K" --output-format csv -d $O/pmc/p3 -o p3 -- $B > $O/pmc/p3.log 2>&1; echo "pass 3 rc=$?"
find $O -name "*.db" -size +20M -delete
ls $O/pmc/p1 $O/pmc/p3 2>/dev/null | head
