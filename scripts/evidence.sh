#!/bin/bash
# The evidence of HEAD on the GPU box, one gpurun call.     usage: scripts/evidence.sh <tag> ["<mode> ..."]
#   full (default)  GPU suite; bench lines (default = BASELINE config 2 with CPU baseline and end-to-end legs; one batch
#                   at a time; 400 steps; dmel; dmel at 2048 piles; arab; the 8(f) paths); rocprofv3 kernel traces
#                   (pipelined, one batch at a time); the counter passes; the long end-to-end stream
#   suite           the GPU suite only
#   quick           bench lines and kernel traces only
#   pmc             the counter passes only
#   e2e             the long end-to-end stream only
#   ab              the kernels' own times of the builds under gpurun_variants/ against the tree's
# Counter passes: ONE kernel per pass (--kernel-include-regex; unrestricted wait / LDS passes hung rocprofv3 on this
# pool in round 5), three SQ_* sets per kernel, FETCH_SIZE / WRITE_SIZE for the alignment kernel (with k_pack, whose
# traffic is known exactly: the calibration) and for k_links2, k_tags, k_chain; a pass that times out is repeated once.
# Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
TAG=${1:-ev}; MODE=${2:-full}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O/pmc
cd $R
# (MODE may list several steps: "suite quick ab"; `full` is all of them)
want() { local m; for m in $MODE; do case " $1 " in *" $m "*) return 0;; esac; done; return 1; }
show() { cut -c1-${2:-200} $1; echo; }

if want "full suite"; then
  ( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
fi
if want "full quick"; then
  timeout 900 python bench.py > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; show $O/bench_ecoli.json.txt
  timeout 300 python bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_one_batch_at_a_time.json.txt 2> /dev/null; show $O/bench_ecoli_one_batch_at_a_time.json.txt 160
  timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_400_steps.json.txt 2> /dev/null; show $O/bench_ecoli_400_steps.json.txt 160
  # (two resident batches instead of three: the round-5 review's question whether the third one earns its 6.8 GB)
  timeout 300 python bench.py --in-flight 2 --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_in_flight_2.json.txt 2> /dev/null; show $O/bench_ecoli_in_flight_2.json.txt 160
  for w in dmel arab; do
    timeout 600 python bench.py --workload $w --no-cpu-baseline --no-end-to-end > $O/bench_$w.json.txt 2> $O/bench_$w.err; show $O/bench_$w.json.txt 160
  done
  timeout 600 python bench.py --workload dmel --piles 2048 --no-cpu-baseline --no-end-to-end > $O/bench_dmel_2048.json.txt 2> $O/bench_dmel_2048.err; show $O/bench_dmel_2048.json.txt 160
  for w in trim align1500 utg; do
    timeout 300 python bench.py --workload $w --steps 5 --warmup 2 > $O/bench_$w.json.txt 2> $O/bench_$w.err; show $O/bench_$w.json.txt 160
  done
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $O/kt.log 2>&1
  python $R/scripts/rocpd_summary.py $(ls $O/kt/*/*.db $O/kt/*.db 2>/dev/null | head -1) > $O/kernel_stats_pipelined.txt 2>&1; head -12 $O/kernel_stats_pipelined.txt | cut -c1-140
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kts -o kts -- python $R/bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/kts.log 2>&1
  python $R/scripts/rocpd_summary.py $(ls $O/kts/*/*.db $O/kts/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt | cut -c1-140
  cd $R
fi
if want "full ab"; then
  # the builds under gpurun_variants/ (scripts/build_variant.sh) against the tree's: the kernels' own times
  V=$(ls $R/gpurun_variants 2>/dev/null | tr '\n' ' ')
  if [ -n "$V" ]; then $R/scripts/ab_times.sh $TAG/ab "head $V" > $O/ab_times.txt 2>&1; cat $O/ab_times.txt; fi
fi
if want "full e2e"; then
  timeout 1200 python bench.py --workload e2e-long > $O/e2e_long.txt 2> $O/e2e_long.err; cat $O/e2e_long.txt
fi
if want "full pmc"; then
  cd /tmp && export TMPDIR=/tmp
  export FALCON_AMD_DEVICE_PACK=1   # (the batch staged through k_pack: the calibration kernel of the traffic passes)
  B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
  P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS"
  P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"
  P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
  pass() {  # <dir> <name> <kernel regex> <counters...>: one pass, repeated once when it times out
    local d=$1 n=$2 k=$3; shift 3
    for try in 1 2; do
      rm -rf $d/$n
      timeout 240 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$k" --output-format csv -d $d/$n -o $n -- $B > $d/$n.log 2>&1
      rc=$?; [ $rc -ne 124 ] && break
    done
    echo "pmc $n rc=$rc"
  }
  for k in k_align2 k_chain k_links2 k_score2 k_tags k_backtrace k_seed_index; do
    for i in 1 2 3; do eval ctrs=\$P$i; pass $O/pmc ${k}_p$i "^$k" $ctrs; done
  done
  python $R/scripts/pmc_table.py $O/pmc > $O/pmc_table.txt 2>&1
  python $R/scripts/pmc_issue_record.py $O/pmc k_align ecoli > $O/pmc_issue.txt 2>&1; tail -3 $O/pmc_issue.txt
  mkdir -p $O/pmc_ecoli $O/pmc_other
  for c in FETCH_SIZE WRITE_SIZE; do pass $O/pmc_ecoli $c "k_align|k_pack" $c; done
  python $R/scripts/pmc_traffic_record.py $O/pmc_ecoli k_align ecoli 2.0 > $O/pmc_traffic_ecoli.txt 2>&1; tail -4 $O/pmc_traffic_ecoli.txt
  for k in k_links2 k_tags k_chain; do
    for c in FETCH_SIZE WRITE_SIZE; do pass $O/pmc_other ${k}_$c "^$k|k_pack" $c; done
  done
  python $R/scripts/pmc_table.py $O/pmc_other > $O/pmc_traffic_other.txt 2>&1
  cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json; cp $R/profiles/pmc_issue.json $O/pmc_issue.json
  unset FALCON_AMD_DEVICE_PACK
  find $O -name "*.db" -size +5M -delete
  find $O -name "*.csv" -size +2M -delete
  cd $R
  # the default line again, now that the traffic and issue records of THIS build are on file
  timeout 400 python bench.py --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_with_records.json.txt 2> /dev/null; show $O/bench_ecoli_with_records.json.txt
fi
find $O -name "*.db" -size +5M -delete
