#!/bin/bash
# Full GPU test suite without stopping at the first failure; log -> gpurun_out/<TAG>_tests.log
# usage: gpu_tests.sh TAG [pytest args...]
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=25 --durations=15 "$@" > gpurun_out/${TAG}_tests.log 2>&1
tail -40 gpurun_out/${TAG}_tests.log
