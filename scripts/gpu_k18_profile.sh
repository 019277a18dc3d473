#!/bin/bash
# K18 under rocprofv3: the two kernels' durations on the 2 GB round-trip test
TAG=${1:-r05k18prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o k18 -- python -m pytest $R/tests/test_gpu_parity.py -x -q -s -k "bgzf_deflate_at_size" > $OUT/run.log 2>&1; echo "rc=$?"
grep -E "K18 at size|passed|failed" $OUT/run.log
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
cp $f $OUT/k18_kernel_stats.csv
find $OUT/prof -name "*.db" -delete; find $OUT/prof -name "*kernel_trace.csv" -delete
