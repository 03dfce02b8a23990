#!/bin/bash
# usage: scripts/r04_exp7.sh <tag> : the worker end to end (30720 piles), batch-size ramps; every setting twice
TAG=${1:-r04r}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1200 python scripts/exp_e2e.py 3072 10 FALCON_AMD_BATCH_RAMP=6:2 FALCON_AMD_BATCH_RAMP=10:2 FALCON_AMD_BATCH_RAMP=6:1.5 FALCON_AMD_BATCH_RAMP=4:3 FALCON_AMD_NOTHING=1 \
    FALCON_AMD_BATCH_RAMP=6:2 FALCON_AMD_BATCH_RAMP=10:2 FALCON_AMD_BATCH_RAMP=6:1.5 FALCON_AMD_BATCH_RAMP=4:3 FALCON_AMD_NOTHING=2 > $O/e2e.txt 2>&1; cat $O/e2e.txt | cut -c1-210
