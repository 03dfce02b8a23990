#!/bin/bash
# Round-5 evidence of HEAD on the GPU box (one gpurun call).  usage: scripts/r05_evidence.sh <tag> [quick|final]
#   quick: without the GPU tests.  final: the alignment kernel's source is the one the traffic record on file was taken
#   on -- no FETCH_SIZE / WRITE_SIZE passes, dmel / arab without their CPU and end-to-end legs, and of the SQ_* passes
#   only the instruction counts (the two wait / LDS passes hung under rocprofv3 on the r05e2 box: 2 x 420 s).
#   bench lines (default = BASELINE config 2 with CPU baseline and the end-to-end legs; one batch at a time; dmel; arab;
#   the 8(f) paths), rocprofv3 kernel traces (pipelined and one batch at a time), the SQ_* counter passes of every kernel of
#   the path, FETCH_SIZE / WRITE_SIZE passes of the alignment kernel for the three workloads, a 400-step line.
TAG=${1:-r05e}; QUICK=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O/pmc
cd $R
FINAL=""; if [ "$QUICK" = "final" ]; then FINAL=1; QUICK=""; fi
if [ -z "$QUICK" ]; then ( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt; fi
timeout 900 python bench.py > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; cut -c1-200 $O/bench_ecoli.json.txt
timeout 300 python bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_serial.json.txt 2> /dev/null; cut -c1-160 $O/bench_ecoli_serial.json.txt
timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_400_steps.json.txt 2> /dev/null; cut -c1-160 $O/bench_ecoli_400_steps.json.txt
for w in dmel arab; do
  timeout 600 python bench.py --workload $w ${FINAL:+--no-cpu-baseline --no-end-to-end} > $O/bench_$w.json.txt 2> $O/bench_$w.err; cut -c1-160 $O/bench_$w.json.txt
done
for w in trim align1500 utg; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 > $O/bench_$w.json.txt 2> $O/bench_$w.err; cut -c1-160 $O/bench_$w.json.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $O/kt.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kt/*/*.db $O/kt/*.db 2>/dev/null | head -1) > $O/kernel_stats_pipelined.txt 2>&1; head -12 $O/kernel_stats_pipelined.txt | cut -c1-140
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kts -o kts -- python $R/bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/kts.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kts/*/*.db $O/kts/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt | cut -c1-140
# (the counter passes stage the batch through k_pack, whose traffic is known exactly: the calibration)
export FALCON_AMD_DEVICE_PACK=1
for w in $([ -z "$FINAL" ] && echo ecoli dmel arab); do
  B="python $R/bench.py --workload $w --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
  mkdir -p $O/pmc_$w
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 420 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_align|k_pack" --output-format csv -d $O/pmc_$w/$c -o $c -- $B > $O/pmc_$w/$c.log 2>&1; echo "pmc $w $c rc=$?"
  done
  python $R/scripts/pmc_traffic_record.py $O/pmc_$w k_align $w 2.0 > $O/pmc_traffic_$w.txt 2>&1; tail -4 $O/pmc_traffic_$w.txt
done
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
for i in $([ -z "$FINAL" ] && echo 1 2) 3; do
  eval ctrs=\$P$i
  timeout 420 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "k_align|k_links|k_score|k_tags|k_sscan|k_chain|k_backtrace|k_seed_index|k_pack" --output-format csv -d $O/pmc/p$i -o p$i -- $B > $O/pmc/p$i.log 2>&1; echo "pmc pass $i rc=$?"
done
python $R/scripts/pmc_table.py $O/pmc > $O/pmc_table.txt 2>&1
python $R/scripts/pmc_issue_record.py $O/pmc k_align ecoli > $O/pmc_issue.txt 2>&1; tail -3 $O/pmc_issue.txt
cp $R/profiles/pmc_issue.json $O/pmc_issue.json
# what FETCH_SIZE counts for the read shapes of the alignment kernel (scripts/ubench/fetch_shapes.hip)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_shapes -o fs -- $R/scripts/ubench/fetch_shapes.bin > $O/fetch_shapes.log 2>&1
python - <<PY > $O/fetch_shapes.txt 2>&1
import csv, glob
known = {"shape_stream16": 4 << 30, "shape_dword_s16": 4 << 30, "shape_dwordx2": 4 << 30, "shape_gather4": 4 << 30}
print(open("$O/fetch_shapes.log").read().strip().splitlines()[-1])
for f in glob.glob("$O/fetch_shapes/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"].split("(")[0]
        if n in known and row["Counter_Name"] == "FETCH_SIZE":
            kib = float(row["Counter_Value"])
            print("%-16s FETCH_SIZE %12.0f KiB = %6.3f GB counted; %5.3f GB of lines touched -> bytes per counted byte %.3f"
                  % (n, kib, kib * 1024 / 1e9, known[n] / 1e9, known[n] / (kib * 1024)))
PY
cat $O/fetch_shapes.txt
find $O -name "*.db" -size +5M -delete
find $O -name "*.csv" -size +2M -delete
unset FALCON_AMD_DEVICE_PACK
cd $R
# the default line again, now that the traffic and issue records of THIS build are on file
if [ -z "$FINAL" ]; then timeout 400 python bench.py --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_with_records.json.txt 2> /dev/null; cut -c1-200 $O/bench_ecoli_with_records.json.txt; fi
