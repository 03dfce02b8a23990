#!/bin/bash
# usage: scripts/r04_e2e_timeline.sh <tag> : the CLI tests, then the worker's own timeline on a 30720-pile stream
TAG=${1:-r04t}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_parity.py -m gpu -x -q -k "cli or printers or deep or hand_back or fallback" 2>&1 | tail -8 ) > $O/pytest.txt; cat $O/pytest.txt
FALCON_AMD_TIMING=1 timeout 900 python scripts/exp_e2e.py 3072 10 FALCON_AMD_NOTHING=1 > $O/e2e.txt 2>&1; cat $O/e2e.txt
grep -v "DEBUG:.*printer\|DEBUG:.*ingest\|DEBUG:.*stager\|DEBUG:.*runner" /tmp/e2e_stream.txt.err | head -60 > $O/e2e_timeline_other.txt
head -40 /tmp/e2e_stream.txt.err > $O/e2e_timeline_head.txt; tail -30 /tmp/e2e_stream.txt.err > $O/e2e_timeline_tail.txt
cat $O/e2e_timeline_head.txt | cut -c1-200; echo ...; cat $O/e2e_timeline_tail.txt | cut -c1-200
