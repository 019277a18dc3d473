#!/bin/bash
# round 5, call 15: the MAF walks as grids of resident waves that ask for the next pair's offsets (scalar registers) before
# they walk the current pair, against one wave per pair
TAG=${1:-r05n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "maf" -p no:cacheprovider 2>&1 | tail -2
for v in product mafp0 mafp1b8 product; do
  if [ "$v" != product ]; then export WGA_LIB=$R/build_variants/libwgahip_$v.so; else unset WGA_LIB; fi
  echo "== $v"
  timeout 300 python scripts/gpu_maf_kernels.py 2>&1 | grep -E "^K[34]|blocks"
done 2>&1 | tee $OUT/maf_variants.txt
