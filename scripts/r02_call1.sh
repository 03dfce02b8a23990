#!/bin/bash
# Round 2, first GPU call: parity (suite + the whole differential campaign through the HIP
# path), the evidence of HEAD (bench line, kernel-trace stats, PMC passes at the bench's
# batch size incl. the read side of the HBM traffic).  Every line bounded by its own timeout.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -x -q                          > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 300 python scripts/gpu_differential_campaign.py piles 0 72        > $O/campaign_piles.txt 2>&1; tail -1 $O/campaign_piles.txt | cut -c1-400
timeout 200 python scripts/gpu_differential_campaign.py pairs 0 64        > $O/campaign_pairs.txt 2>&1; tail -1 $O/campaign_pairs.txt | cut -c1-400
timeout 300 python bench.py > $O/bench.json.txt 2> $O/bench.err; cut -c1-300 $O/bench.json.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $O/kt.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kt/*/*.db $O/kt/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt | cut -c1-150
pass() { # name timeout extra-args... -- counters
  n=$1; t=$2; shift 2
  timeout $t rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$n -o $n -- python $R/bench.py --steps 1 --warmup 0 --piles 3072 --no-cpu-baseline --no-end-to-end > $O/pmc/$n.log 2>&1
  echo "pmc pass $n rc=$?"
}
mkdir -p $O/pmc
pass p1 150 WRITE_SIZE
pass p2 150 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass p3 150 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass p4 150 TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum
pass p5 150 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass p6 150 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
# the read-side counter that hung in round 1: once more, restricted to k_align's dispatches
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex k_align --output-format csv -d $O/pmc/p7 -o p7 -- python $R/bench.py --steps 1 --warmup 0 --piles 768 --no-cpu-baseline --no-end-to-end > $O/pmc/p7.log 2>&1
echo "pmc pass p7 (FETCH_SIZE, k_align only, 768 piles) rc=$?"
python $R/scripts/pmc_table.py $O/pmc > $O/pmc_table.txt 2>&1; grep -c . $O/pmc_table.txt
# keep the CSVs small: only the counter tables travel back
find $O -name "*.db" -size +20M -delete
