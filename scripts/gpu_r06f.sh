#!/bin/bash
# round 6: the north-star stream (10 M records x mean 50 kop) on this round's tree
TAG=${1:-r06f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 2700 python bench.py --north-star --ns-records 10000000 > $OUT/north_star_10M.json 2> $OUT/north_star.err
echo "rc=$?"; tail -c 1500 $OUT/north_star_10M.json; tail -3 $OUT/north_star.err
