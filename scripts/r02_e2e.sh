#!/bin/bash
# End-to-end worker rate on a long stream, with the per-phase timeline of the last run.
#   bash scripts/r02_e2e.sh <tag> [ENV=VALUE ...]
tag=${1:-e2e}; shift
out=gpurun_out/$tag
mkdir -p "$out"
export FALCON_AMD_TIMING=1
timeout 900 python scripts/exp_e2e.py 3072 3 "$@" > "$out/e2e.txt" 2>&1
cp /tmp/e2e_stream.txt.err "$out/e2e_last.err" 2>/dev/null
tail -5 "$out/e2e.txt"
