#!/bin/bash
# GPU tests (all), then the K5 scaling diagnostic
TAG=${1:-r03g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "config [45]|passed|failed|Error|error" $OUT/pytest_gpu.log | tail -12
timeout 600 python scripts/gpu_k5_scaling.py 1 2 4 > $OUT/k5_scaling.log 2>&1; echo "k5 rc=$?"; cat $OUT/k5_scaling.log | tail -8
