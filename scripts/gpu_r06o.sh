#!/bin/bash
TAG=${1:-r06o}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python scripts/gpu_e2e_at_size.py 2>&1 | grep -v amdgpu.ids | cut -c1-600 | grep -A1 "^maf2paf\|^stat_maf\|^call_maf " | tee $OUT/e2e.txt
