#!/bin/bash
# What to run first with the next GPU budget (each line bounded by its own timeout; about
# 4 GPU-minutes in all): parity, the wide differential campaign, the evidence of HEAD, the
# two open measurements of the worker.  Outputs under gpurun_out/next/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/next; mkdir -p $O
cd $R
timeout 200 python -m pytest tests -m gpu -x -q                       > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 120 python scripts/gpu_differential_campaign.py 0 8           > $O/campaign.txt 2>&1; tail -1 $O/campaign.txt
timeout 500 bash scripts/collect_profiles.sh next/evidence            > $O/collect.txt 2>&1; tail -3 $O/collect.txt
timeout 60  python scripts/exp_batch_policy.py 3072                   > $O/batch_policy.txt 2>&1; cat $O/batch_policy.txt
timeout 90  python scripts/exp_multistream.py 3072 4                  > $O/multistream.txt 2>&1; cat $O/multistream.txt
