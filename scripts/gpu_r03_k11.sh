#!/bin/bash
# K11 fill with the block's two record searches by whole waves: parity, then the rates
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03H}
mkdir -p $OUT
cd $R
timeout 120 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -x -q -m gpu -k "bridge or chain" 2>&1 | tail -2
timeout 200 python scripts/gpu_other_kernels.py 100000 5000 2>&1 | grep -E "^K11" | tee $OUT/k11.log
