#!/bin/bash
# round 4: the streaming row kernel — correctness at size, then variants on the same buffers
TAG=${1:-r04c}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python scripts/gpu_row_kernel_check.py 3 > $OUT/check3.log 2>&1; echo "check rc=$?"; tail -7 $OUT/check3.log
timeout 900 python scripts/gpu_k2_same_buffers.py run "$@" > $OUT/same_buffers.log 2>&1; echo "ab rc=$?"; grep -v amdgpu $OUT/same_buffers.log | tail -40
