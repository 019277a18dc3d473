#!/bin/bash
# round 5, call 8: K5's list pass as a grid of resident waves WITHOUT the early request of the next tile (same registers as one
# wave per tile: stores need no acknowledgement before the wave goes on), three register budgets; the product build beside them
TAG=${1:-r05h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for v in "" k5persist k5persist6 k5persist7; do
  echo "== ${v:-product}"
  if [ -n "$v" ]; then export WGA_LIB=$R/build_variants/libwgahip_$v.so; else unset WGA_LIB; fi
  K5_MODE=both K5_REPS=3 timeout 600 python scripts/gpu_k5_scaling.py 10 2>&1 | grep -E "^chunks|fused ==" | tail -3
done 2>&1 | tee $OUT/k5_persist.txt
