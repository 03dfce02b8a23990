#!/bin/bash
# Round-2 evidence of HEAD on the GPU box: parity, bench lines (three workloads, pipelined and
# one batch at a time), rocprofv3 kernel stats, PMC passes (HBM traffic both sides, issue/wait,
# LDS bank conflicts), secondary entry points.  usage: scripts/r02_evidence.sh <tag>
TAG=${1:-r02e}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O/pmc
cd $R
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 200 python scripts/gpu_differential_campaign.py piles 0 300 > $O/campaign.txt 2>&1
timeout 200 python scripts/gpu_differential_campaign.py pairs 0 256 >> $O/campaign.txt 2>&1; cut -c1-160 $O/campaign.txt
timeout 600 python bench.py > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; cut -c1-160 $O/bench_ecoli.json.txt
timeout 300 python bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_serial.json.txt 2> /dev/null; cut -c1-160 $O/bench_ecoli_serial.json.txt
for w in dmel arab; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json.txt 2> $O/bench_$w.err; cut -c1-160 $O/bench_$w.json.txt
done
timeout 200 python scripts/exp_secondary.py > $O/secondary.txt 2>&1; tail -12 $O/secondary.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $O/kt.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kt/*/*.db $O/kt/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kts -o kts -- python $R/bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/kts.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kts/*/*.db $O/kts/*.db 2>/dev/null | head -1) > $O/kernel_stats_serial.txt 2>&1; head -14 $O/kernel_stats_serial.txt | cut -c1-140
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_align|k_pack" --output-format csv -d $O/pmc/$c -o $c -- $B > $O/pmc/$c.log 2>&1; echo "pmc $c rc=$?"
done
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  # (restricted to the kernels of the path: unrestricted passes hang rocprofv3 on this pool every other time)
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "k_align|k_links|k_score|k_tags|k_chain|k_backtrace|k_seed_index|k_pack" --output-format csv -d $O/pmc/p$i -o p$i -- $B > $O/pmc/p$i.log 2>&1; echo "pmc pass $i rc=$?"
done
python $R/scripts/pmc_table.py $O/pmc > $O/pmc_table.txt 2>&1
find $O -name "*.db" -size +20M -delete
# kernel timeline of pipelined steps (every dispatch with its queue, duration, gaps)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -o tl -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/tl.log 2>&1
python $R/scripts/timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) $(find $O/tl -name "*memory_copy_trace.csv" | head -1) > $O/timeline_pipelined.txt 2>&1; head -18 $O/timeline_pipelined.txt
# the worker end to end on a long stream (text in -> FASTA out, one process)
cd $R
FALCON_AMD_TIMING=1 timeout 600 python scripts/exp_e2e.py 3072 3 FALCON_AMD_NOTHING=1 > $O/e2e.txt 2>&1; tail -2 $O/e2e.txt
cp /tmp/e2e_stream.txt.err $O/e2e_last.err 2>/dev/null
# instruction counts by class: restricted to the kernels of the path (an unrestricted pass hung once)
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-include-regex "k_align|k_links|k_score|k_tags|k_chain|k_backtrace|k_seed_index" --output-format csv -d $O/pmc/p9 -o p9 -- python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end > $O/pmc/p9.log 2>&1; echo "pmc insts rc=$?"
python $R/scripts/pmc_table.py $O/pmc > $O/pmc_table.txt 2>&1
