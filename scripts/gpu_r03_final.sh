#!/bin/bash
# the round's evidence in one call: smoke, the driver's bench command plain and under rocprofv3 --kernel-trace --stats, the PMC
# passes, the stated-size configs, the RCCL branches at world size 1.   usage: gpurun -- 'bash scripts/gpu_r03_final.sh r03'
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== rocm-smi" > $OUT/env.log; rocm-smi --showproductname >> $OUT/env.log 2>&1; nproc >> $OUT/env.log; lscpu | head -20 >> $OUT/env.log; free -g >> $OUT/env.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench (the driver's command)"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
echo "== the same command under rocprofv3 --kernel-trace --stats"
cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $R; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do echo $f; head -8 $f | cut -c1-160; done
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
echo "== PMC passes"; bash scripts/gpu_pmc.sh ${TAG}_pmc "sq1 sq2 sq3 fetch write tcc" > $OUT/pmc.log 2>&1; tail -8 $OUT/pmc.log | cut -c1-400
echo "== stated-size configs"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "stated_size or config2_full or config3_full" > $OUT/full_configs.txt 2>&1; echo "rc=$?"; grep -E "config|passed|failed" $OUT/full_configs.txt | tail -12
echo "== RCCL branches at world size 1"
timeout 600 python bench.py --gpus 1 --steps 3 --warmup 5 --force-dist --no-cpu-baseline --no-extras --no-e2e --check 4 > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err; echo "bench --force-dist rc=$?"; tail -2 $OUT/bench_force_dist.err
WGA_DIST_FORCE=1 timeout 600 python -m pytest tests/test_gpu_cli.py -q -m gpu -k "dist_cli_one_rank" > $OUT/dist_cli_nccl_world1.log 2>&1; echo "dist_cli (WGA_DIST_FORCE=1) rc=$?"; tail -2 $OUT/dist_cli_nccl_world1.log
