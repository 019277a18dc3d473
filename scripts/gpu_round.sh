#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything lands in
# gpurun_out/ (merged back by gpurun).  Usage: gpurun -- 'bash scripts/gpu_round.sh [tag]'
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== rocm-smi" > $OUT/env.log; rocm-smi --showproductname >> $OUT/env.log 2>&1; nproc >> $OUT/env.log; lscpu | head -20 >> $OUT/env.log; free -g >> $OUT/env.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -25 $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprofv3 kernel stats"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-extras --no-placement-probe --check 0 > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $R; find $OUT/prof -name '*stats*' | head; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -12 $f; done
# keep the merge small: drop the raw per-dispatch trace if it is big
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
