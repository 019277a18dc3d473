#!/bin/bash
# round 5, call 5: what a read-only kernel reaches on this part (micro), what K1's atomics cost (A/B build)
TAG=${1:-r05e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
echo "== micro: read_only"
( cd scripts/micro && timeout 300 ./read_only ) 2>&1 | tee $OUT/read_only.txt
echo "== K1"
timeout 300 python scripts/gpu_k1.py k1noatom 2>&1 | grep "^K1" | tee $OUT/k1.txt
