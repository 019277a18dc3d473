#!/bin/bash
# round 5, call 10: the command line file to file at size after the call pipeline change; the whole GPU suite on the final tree
TAG=${1:-r05j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python scripts/gpu_e2e_at_size.py > $OUT/e2e_at_size.log 2>&1; echo "e2e rc=$?"; grep -A1 -E "^call_maf|^stat_maf|^maf2paf|^pafcov|^stat_paf" $OUT/e2e_at_size.log | cut -c1-420
timeout 1800 python -m pytest tests -x -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "config 4 at size|passed|failed" $OUT/pytest_gpu.log | tail -5
