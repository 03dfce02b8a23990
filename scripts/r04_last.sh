#!/bin/bash
# usage: scripts/r04_last.sh <tag> : the counter pass that hung in the evidence call, the default bench line and the worker end to end once more
TAG=${1:-r04g}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O/pmc; cd $R
( timeout 300 python -m pytest tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3 ) > $O/pytest_cli.txt; cat $O/pytest_cli.txt
timeout 120 python scripts/exp_e2e.py 3072 10 FALCON_AMD_NOTHING=1 FALCON_AMD_NOTHING=2 > $O/e2e.txt 2>&1; cat $O/e2e.txt | cut -c1-200
timeout 400 python bench.py > $O/bench_ecoli.json.txt 2> $O/bench_ecoli.err; cut -c1-200 $O/bench_ecoli.json.txt
cd /tmp && export TMPDIR=/tmp
export FALCON_AMD_DEVICE_PACK=1
B="python $R/bench.py --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-end-to-end"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-include-regex "k_align|k_links|k_score|k_tags|k_sscan|k_chain|k_backtrace|k_seed_index|k_pack" --output-format csv -d $O/pmc/p1 -o p1 -- $B > $O/pmc/p1.log 2>&1; echo "pmc pass 1 rc=$?"
find $O -name "*.db" -size +5M -delete; find $O -name "*.csv" -size +2M -delete
