#!/bin/bash
# Final evidence of round 3 in one short call: bench line (default), traffic counters, kernel
# traces, the one-batch-at-a-time line, dmel / arab lines without their CPU legs.
TAG=${1:-r03f}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O/pmc
cd $R
timeout 400 python bench.py > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; cut -c1-160 $O/bench_ecoli.json.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
for c in WRITE_SIZE FETCH_SIZE; do
  FALCON_AMD_DEVICE_PACK=1 timeout 120 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_align|k_pack" --output-format csv -d $O/pmc/$c -o $c -- $B > $O/pmc/$c.log 2>&1; echo "pmc $c rc=$?"
done
python $R/scripts/pmc_traffic_record.py $O/pmc k_align ecoli 1.0 > $O/pmc_traffic.txt 2>&1; tail -12 $O/pmc_traffic.txt | head -8
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $O/kt.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kt/*/*.db $O/kt/*.db 2>/dev/null | head -1) > $O/kernel_stats_pipelined.txt 2>&1; head -4 $O/kernel_stats_pipelined.txt | cut -c1-130
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kts -o kts -- python $R/bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/kts.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kts/*/*.db $O/kts/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt 2>&1; head -4 $O/kernel_stats.txt | cut -c1-130
cd $R
timeout 200 python bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_serial.json.txt 2> /dev/null; cut -c1-130 $O/bench_ecoli_serial.json.txt
for w in dmel arab; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline --no-end-to-end > $O/bench_$w.json.txt 2> $O/bench_$w.err; cut -c1-130 $O/bench_$w.json.txt
done
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-include-regex "k_align" --output-format csv -d $O/pmc/p3 -o p3 -- $B > $O/pmc/p3.log 2>&1; echo "pmc insts rc=$?"
python $R/scripts/pmc_table.py $O/pmc > $O/pmc_table.txt 2>&1
find $O -name "*.db" -size +5M -delete
find $O -name "*.csv" -size +2M -delete
