#!/bin/bash
# Round 2, second GPU call: issue-rate microbenchmark, the self-checking bench line of the three
# workloads, kernel stats of dmel/arab, FETCH_SIZE/WRITE_SIZE of k_align with k_pack as calibration.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02b; mkdir -p $O/pmc
cd $R
(nproc; lscpu | grep -i "model name\|socket\|core(s)\|thread"; free -g | head -2) > $O/host.txt 2>&1
timeout 120 scripts/ubench/issue_rates.bin > $O/issue_rates.txt 2>&1; tail -25 $O/issue_rates.txt
timeout 400 python bench.py > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; cut -c1-200 $O/bench_ecoli.json.txt; tail -3 $O/bench_ecoli.err
for w in dmel arab; do
  timeout 400 python bench.py --workload $w > $O/bench_$w.json.txt 2> $O/bench_$w.err; cut -c1-200 $O/bench_$w.json.txt; tail -3 $O/bench_$w.err
done
cd /tmp && export TMPDIR=/tmp
for w in dmel arab; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python $R/bench.py --workload $w --no-cpu-baseline --no-end-to-end > $O/kt_$w.log 2>&1
  python $R/scripts/rocpd_summary.py $(ls $O/kt_$w/*/*.db $O/kt_$w/*.db 2>/dev/null | head -1) > $O/kernel_stats_$w.txt 2>&1; head -12 $O/kernel_stats_$w.txt | cut -c1-150
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_align|k_pack" --output-format csv -d $O/pmc/$c -o $c -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $O/pmc/$c.log 2>&1
  echo "pmc $c rc=$?"
done
find $O -name "*.db" -size +20M -delete
