#!/bin/bash
# usage: scripts/r04_exp4.sh <tag> : the worker's timeline on the 30720-pile stream, three runs; counters of the rewritten kernels
TAG=${1:-r04n}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O/pmc; cd $R
FALCON_AMD_TIMING=1 timeout 600 python scripts/exp_e2e.py 3072 10 FALCON_AMD_NOTHING=1 FALCON_AMD_NOTHING=2 > $O/e2e.txt 2>&1; cat $O/e2e.txt | cut -c1-200
grep -v "printer:\|ingest:\|stager:\|runner:\|fa_batch_create\|msa stage\|align launch\|fa_batch_submit" /tmp/e2e_stream.txt.err | head -40 | cut -c1-250 > $O/e2e_other.txt; cat $O/e2e_other.txt
head -60 /tmp/e2e_stream.txt.err > $O/e2e_timeline_head.txt; tail -30 /tmp/e2e_stream.txt.err > $O/e2e_timeline_tail.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "k_links|k_score|k_tags|k_chain|k_backtrace|k_seed_index" --output-format csv -d $O/pmc/p$i -o p$i -- $B > $O/pmc/p$i.log 2>&1; echo "pmc pass $i rc=$?"
done
python $R/scripts/pmc_table.py $O/pmc > $O/pmc_table.txt 2>&1
find $O -name "*.db" -size +5M -delete; find $O -name "*.csv" -size +2M -delete
grep -A30 "^k_backtrace\|^k_tags\|^k_chain" $O/pmc_table.txt | grep -E "^k_|INSTS_(VALU|SALU|LDS)|LDS_(BANK|IDX)|WAIT_ANY|WAIT_INST_ANY|WAVE_CYCLES|->" | head -60
