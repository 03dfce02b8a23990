#!/bin/bash
# A/B of library builds under gpurun_variants/<name>/libfalcon_amd.so (and "head" = the tree's own):
# k_align2's time alone (3 unpipelined steps) and, with "pmc", its instruction / wait counters.
# usage: scripts/ab_counters.sh <tag> "<variants>" [pmc]
TAG=${1:-r06ab}; VARS=${2:-head}; PMC=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FALCON_AMD_DEVICE_PACK=1
for v in $VARS; do
  if [ "$v" = "head" ]; then unset FALCON_AMD_LIB; else export FALCON_AMD_LIB=$R/gpurun_variants/$v/libfalcon_amd.so; fi
  timeout 300 python $R/bench.py --no-pipeline --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_$v.json.txt 2> $O/bench_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$v.json.txt").read().strip().splitlines()[-1])
    a = d["align"]
    print("%-10s k_align %.3f ms  step %.2f ms  value %.1f M  replacements %d parkings %d placements %d single %.1f %%" % ("$v", d["kernel_ms"]["k_align"], d["ms_per_step"], d["value"] / 1e6,
          a["replacements_in_loop"], a["parkings"], a["placements"], 100.0 * a["single_iterations"] / (a["single_iterations"] + a["pair_iterations"])))
except Exception as e:
    print("$v: bench line unreadable:", e, open("$O/bench_$v.err").read()[-500:])
PY
  if [ -n "$PMC" ]; then
    B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
    P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES"
    P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
    for i in 3 1; do
      eval ctrs=\$P$i
      timeout 240 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "^k_align2" --output-format csv -d $O/${v}_p$i -o p$i -- $B > $O/${v}_p$i.log 2>&1 || echo "pass $i rc=$?"
    done
    python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in sorted(glob.glob("$O/${v}_p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
print("   ", "  ".join("%s %.4g" % (k.replace("SQ_", ""), v) for k, v in sorted(tot.items())))
PY
  fi
done
find $O -name "*.db" -size +5M -delete
find $O -name "*.csv" -size +2M -delete
