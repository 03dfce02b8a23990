#!/bin/bash
# Round-4 evidence of HEAD on the GPU box.  usage: scripts/r04_evidence.sh <tag> [quick]
TAG=${1:-r04e}; QUICK=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O/pmc
cd $R
if [ -z "$QUICK" ]; then ( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt; fi
timeout 700 python bench.py > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; cut -c1-200 $O/bench_ecoli.json.txt
timeout 300 python bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/bench_ecoli_serial.json.txt 2> /dev/null; cut -c1-160 $O/bench_ecoli_serial.json.txt
FALCON_AMD_LINKS1=1 FALCON_AMD_SCORE1=1 timeout 400 python bench.py --no-cpu-baseline --no-end-to-end --steps 4 > $O/bench_ecoli_links1_score1.json.txt 2> /dev/null; cut -c1-160 $O/bench_ecoli_links1_score1.json.txt
if [ -z "$QUICK" ]; then
for w in dmel arab; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json.txt 2> $O/bench_$w.err; cut -c1-160 $O/bench_$w.json.txt
done
fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $O/kt.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kt/*/*.db $O/kt/*.db 2>/dev/null | head -1) > $O/kernel_stats_pipelined.txt 2>&1; head -14 $O/kernel_stats_pipelined.txt | cut -c1-140
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kts -o kts -- python $R/bench.py --no-pipeline --no-cpu-baseline --no-end-to-end > $O/kts.log 2>&1
python $R/scripts/rocpd_summary.py $(ls $O/kts/*/*.db $O/kts/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt 2>&1; head -14 $O/kernel_stats.txt | cut -c1-140
# (the counter passes stage the batch through k_pack, whose traffic is known exactly: the calibration)
export FALCON_AMD_DEVICE_PACK=1
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_align|k_pack" --output-format csv -d $O/pmc/$c -o $c -- $B > $O/pmc/$c.log 2>&1; echo "pmc $c rc=$?"
done
python $R/scripts/pmc_traffic_record.py $O/pmc k_align ecoli 1.0 > $O/pmc_traffic.txt 2>&1; tail -22 $O/pmc_traffic.txt
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 420 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "k_align|k_links|k_score|k_tags|k_sscan|k_chain|k_backtrace|k_seed_index|k_pack" --output-format csv -d $O/pmc/p$i -o p$i -- $B > $O/pmc/p$i.log 2>&1; echo "pmc pass $i rc=$?"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_tags|k_links2|k_score2|k_backtrace|k_sscan" --output-format csv -d $O/pmc/msa_$c -o msa_$c -- $B > $O/pmc/msa_$c.log 2>&1; echo "pmc msa $c rc=$?"
done
python $R/scripts/pmc_table.py $O/pmc > $O/pmc_table.txt 2>&1
find $O -name "*.db" -size +5M -delete
find $O -name "*.csv" -size +2M -delete
cd $R
unset FALCON_AMD_DEVICE_PACK
FALCON_AMD_TIMING=1 timeout 600 python scripts/exp_e2e.py 3072 10 FALCON_AMD_NOTHING=1 FALCON_AMD_NOTHING=2 > $O/e2e.txt 2>&1
grep -i "steady" /tmp/e2e_stream.txt.err | tail -1 >> $O/e2e.txt
grep -v "printer:\|ingest:\|stager:\|runner:\|fa_batch_create\|msa stage\|align launch\|fa_batch_submit" /tmp/e2e_stream.txt.err | head -30 | cut -c1-250 > $O/e2e_timeline_marks.txt
tail -4 $O/e2e.txt | cut -c1-220
bash scripts/r04_step_vs_batch_size.sh $TAG/batch_size > /dev/null 2>&1; python - <<EOF
import json
out = ["# python bench.py --piles N [--no-pipeline] --no-cpu-baseline --no-end-to-end: step time against batch size, final kernels"]
for n in (473, 946, 1536, 3072):
    for suf in ("", "_serial"):
        try:
            d = json.loads(open("$O/batch_size/bench_%d%s.json.txt" % (n, suf)).read().strip().splitlines()[-1])
            out.append("%5d piles %-10s ms_per_step %6.2f  us_per_pile %5.2f  piles/s %6.0f  kernel_ms %s" % (n, suf[1:] or "pipelined", d["ms_per_step"], 1e3 * d["ms_per_step"] / n, d["piles_per_sec"], d["kernel_ms"]))
        except Exception as e:
            out.append("%d %s unreadable: %r" % (n, suf, e))
open("$O/step_vs_batch_size.txt", "w").write("\n".join(out) + "\n")
EOF
cut -c1-110 $O/step_vs_batch_size.txt
