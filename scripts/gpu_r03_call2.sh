#!/bin/bash
# the window row kernel (expand_variant 2): parity on the GPU, then same-buffer timings of its builds
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python scripts/gpu_k2w_check.py > $OUT/k2w_check.log 2>&1; echo "check rc=$?"; tail -8 $OUT/k2w_check.log
shift
timeout 1200 python scripts/gpu_k2_same_buffers.py run tree:expand_variant=0 tree:expand_variant=2 "$@" \
   --shape 100000,5000,50 --shape 10000,50000,50 --shape 1000000,500,50 --shape 100000,5000,1000 > $OUT/k2w_ab.log 2>&1; echo "ab rc=$?"; cat $OUT/k2w_ab.log
