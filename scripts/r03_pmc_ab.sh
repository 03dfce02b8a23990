#!/bin/bash
# HBM traffic counters of k_align2 for library variants (gpurun_variants/), and with the trace-back off
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03pmcab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
run() { # name env...
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    env "$@" timeout 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_align2" --output-format csv -d $O/$name/$c -o $c -- $B > $O/$name.$c.log 2>&1
  done
  python - $O/$name $name <<'P'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("k_align2"): agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(sys.argv[2], {k:(round(sum(v)/len(v)*1024/1e9,2),len(v)) for k,v in agg.items()}, "GB per launch (raw), dispatches")
P
}
run head FALCON_AMD_LIB=$R/gpurun_variants/libfalcon_amd_head.so
run pol4 FALCON_AMD_LIB=$R/gpurun_variants/libfalcon_amd_pol4.so
run head_notrace FALCON_AMD_LIB=$R/gpurun_variants/libfalcon_amd_head.so FALCON_AMD_A2_DEBUG=1
run head_ring16k FALCON_AMD_LIB=$R/gpurun_variants/libfalcon_amd_head.so FALCON_AMD_RING=16384
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
