#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 ) > gpurun_out/r03_t3_pytest.txt
tail -25 gpurun_out/r03_t3_pytest.txt
