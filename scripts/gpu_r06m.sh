#!/bin/bash
# round 6: the MAF reader's piece size under the new back end (call / stat at 2 M blocks, pieces of 1 GiB .. 128 MiB)
TAG=${1:-r06m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python scripts/gpu_e2e_at_size.py maf-pieces 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee $OUT/maf_pieces.txt
