#!/bin/bash
# K7 / K12 with the next step's ops fetched behind the current step: parity, then the rates on 5 kop records
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03G}
mkdir -p $OUT
cd $R
timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "call or dotplot or event or long_records" 2>&1 | tail -2
timeout 200 python scripts/gpu_other_kernels.py 100000 5000 2>&1 | grep -E "^K7|^K12|^K10" | tee $OUT/k7.log
