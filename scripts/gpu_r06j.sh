#!/bin/bash
# round 6: the MAF stream kernels at 4 / 5 / 6 (product) / 8 blocks per CU (register budgets 128 / 96 / 80 / 64)
TAG=${1:-r06j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for v in product mafb4 mafb5 mafb8 product; do
  if [ "$v" != product ]; then export WGA_LIB=$R/build_variants/libwgahip_$v.so; else unset WGA_LIB; fi
  echo "== $v"
  timeout 120 python scripts/gpu_maf_kernels.py 2>&1 | grep -E "^K[34]"
done 2>&1 | tee $OUT/maf_blocks_per_cu.txt
