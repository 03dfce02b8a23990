#!/bin/bash
# usage: scripts/r04_exp5.sh <tag> : the worker end to end (30720 piles) under a few settings
TAG=${1:-r04p}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt; cat $O/pytest.txt
timeout 1200 python scripts/exp_e2e.py 3072 10 FALCON_AMD_NOTHING=1 FALCON_AMD_SLOW_EXIT=1 FALCON_AMD_ENGINES_PER_DEVICE=2 FALCON_AMD_READ_AHEAD=4 \
   FALCON_AMD_READ_AHEAD=3 FALCON_AMD_BATCH_BASES=300000000 FALCON_AMD_BATCH_BASES=600000000 FALCON_AMD_RUNNERS_PER_ENGINE=4 FALCON_AMD_NOTHING=2 FALCON_AMD_NOTHING=3 > $O/e2e.txt 2>&1; cat $O/e2e.txt | cut -c1-210
