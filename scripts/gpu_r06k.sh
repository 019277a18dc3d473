#!/bin/bash
# round 6: K19 against the oracle, every block, under other rule paths (every indel a row; chunk cuts of a few hundred / a few dozen columns)
TAG=${1:-r06k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for args in "50000 0 1000000" "50000 5 700" "50000 50 64" "30000 1 1"; do
  timeout 600 python scripts/gpu_k19_check.py $args 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300
done | tee $OUT/k19_check.txt
