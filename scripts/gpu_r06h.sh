#!/bin/bash
TAG=${1:-r06h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -25 | cut -c1-500 | tee $OUT/k19_test.txt
