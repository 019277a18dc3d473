#!/bin/bash
# round 6: the N > 1 branches of bench.py on one GPU (process group forced at world size 1; strong-scaling shape), and --gpus 2 on a one-GPU box
TAG=${1:-r06n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python bench.py --force-dist --scaling strong --steps 5 --warmup 2 --no-extras --no-e2e --no-cpu-baseline > $OUT/bench_force_dist.json 2> $OUT/err.txt; echo "force-dist rc=$?"
cut -c1-700 $OUT/bench_force_dist.json; tail -2 $OUT/err.txt
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_gpus2.txt 2>&1; echo "gpus 2 on this box rc=$?"; tail -2 $OUT/bench_gpus2.txt | cut -c1-300
