#!/bin/bash
# usage: scripts/r04_e2e_sweep.sh <tag> : forced hand-back tests, then the worker end to end under a few settings
TAG=${1:-r04s}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hand_back" 2>&1 | tail -6 ) > $O/pytest.txt; cat $O/pytest.txt
timeout 1200 python scripts/exp_e2e.py 3072 10 FALCON_AMD_NOTHING=1 FALCON_AMD_RUNNERS_PER_ENGINE=4 FALCON_AMD_RUNNERS_PER_ENGINE=5 \
   FALCON_AMD_BATCH_BASES=600000000 FALCON_AMD_BATCH_BASES=800000000 FALCON_AMD_READER_SLOTS1=1 FALCON_AMD_READER_SSE2=1 FALCON_AMD_NOTHING=2 > $O/e2e.txt 2>&1; cat $O/e2e.txt | cut -c1-210
