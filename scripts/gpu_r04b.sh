#!/bin/bash
# round 4, call 2: the streaming row kernel (expand_variant 3) on the GPU — oracle battery + byte-for-byte against v1 at size,
# then v1 / window / streaming on the same buffers
TAG=${1:-r04b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python scripts/gpu_k2w_check.py 3 > $OUT/check3.log 2>&1; echo "check rc=$?"; tail -12 $OUT/check3.log
timeout 600 python scripts/gpu_k2_same_buffers.py run tree:expand_variant=0 tree:expand_variant=2 tree:expand_variant=3 tree:expand_variant=3,expand_job_tiles=8 tree:expand_variant=3,expand_job_tiles=2 tree:expand_variant=3,expand_job_tiles=16 \
   --shape 100000,5000,50 --shape 10000,50000,50 --shape 100000,5000,1000 > $OUT/same_buffers.log 2>&1; echo "ab rc=$?"; grep -v amdgpu $OUT/same_buffers.log | tail -30
