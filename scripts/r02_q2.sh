R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/q2; mkdir -p $O; cd $R
timeout 100 scripts/ubench/latency.bin > $O/latency.txt 2>&1; cat $O/latency.txt
timeout 200 python bench.py --no-cpu-baseline --no-end-to-end --piles 768 > $O/b768.txt 2>$O/b768.err; python -c "
import json; d=json.loads(open('$O/b768.txt').readline()); print(768, d['ms_per_step'], d['kernel_ms'])"
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "k_score|k_links" --output-format csv -d $O/p1 -o p1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $O/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-include-regex "k_score|k_links" --output-format csv -d $O/p2 -o p2 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $O/p2.log 2>&1
python $R/scripts/pmc_table.py $O | grep -v "^   TC"
