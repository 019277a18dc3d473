#!/bin/bash
TAG=${1:-r06i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
K5_COLD=2 K5_REPS=1 K5_MODE=fused timeout 600 python scripts/gpu_k5_scaling.py 10 2>&1 | grep -v "^W2\|^E2\|amdgpu.ids" | tee $OUT/k5_cold2.log
