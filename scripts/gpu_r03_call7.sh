#!/bin/bash
# all GPU tests, the op-stream kernels on both record lengths, the command line end to end (20 000 records)
TAG=${1:-r03m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head
timeout 600 python scripts/gpu_other_kernels.py 100000 5000 > $OUT/other_5k.log 2>&1; echo "other 5k rc=$?"; grep -E "^K" $OUT/other_5k.log
timeout 600 python scripts/gpu_other_kernels.py 10000 50000 > $OUT/other_50k.log 2>&1; echo "other 50k rc=$?"; grep -E "^K7|^K10|^K12" $OUT/other_50k.log
timeout 600 python scripts/gpu_maf_kernels.py > $OUT/maf_kernels.log 2>&1; echo "maf rc=$?"; tail -8 $OUT/maf_kernels.log
timeout 900 python scripts/gpu_cli_e2e.py 20000 /tmp/wga_e2e 100000 > $OUT/cli_e2e_20k.txt 2>&1; echo "e2e rc=$?"; grep -v "timing" $OUT/cli_e2e_20k.txt | head -20; grep -A1 "call paf" $OUT/cli_e2e_20k.txt
