#!/bin/bash
TAG=${1:-r06l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -x -k "maf or call" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests.txt
