#!/bin/bash
# bench variants of the two-halves run: back-stream count / priority, batches in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03x2
export TMPDIR=/tmp
run() {  # name, env..., -- bench args
    name=$1; shift
    envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end "$@" > gpurun_out/r03x2/$name.json 2> gpurun_out/r03x2/$name.err
    python - "$name" <<'P'
import json,sys
try:
    r=json.loads(open("gpurun_out/r03x2/%s.json"%sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[1], r["value"], r["ms_per_step"], {k:round(v,1) for k,v in r["kernel_ms"].items()})
except Exception as e:
    print(sys.argv[1], "failed", e)
P
}
run base --
run prio_low FALCON_AMD_BACK_PRIO=low --
run prio_same FALCON_AMD_BACK_PRIO=same --
run nback1 FALCON_AMD_NBACK=1 --
run nback3 FALCON_AMD_NBACK=3 --
run flight4 -- --in-flight 4
run flight2 -- --in-flight 2
