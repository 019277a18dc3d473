#!/bin/bash
# round 3, first call: GPU tests (stated-size configs un-gated), smoke, the driver's bench command, and rocprofv3
# --kernel-trace --stats of that exact command.  usage: gpurun -- 'bash scripts/gpu_r03_call1.sh [tag]'
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== rocm-smi" > $OUT/env.log; rocm-smi --showproductname >> $OUT/env.log 2>&1; nproc >> $OUT/env.log; lscpu | head -20 >> $OUT/env.log; free -g >> $OUT/env.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; grep -E "config [45]|passed|failed" $OUT/pytest_gpu.log | tail -8
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== bench (driver command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprofv3 kernel stats of the same command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $R; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -14 $f; done
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
