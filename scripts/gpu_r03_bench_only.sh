#!/bin/bash
# the driver's bench command plain and under rocprofv3 --kernel-trace --stats (the two files of profiles/r03_bench_n1*.json)
TAG=${1:-r03w}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $R; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -4 $f | cut -c1-140; done
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
