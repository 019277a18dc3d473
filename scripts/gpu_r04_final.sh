#!/bin/bash
# round 4 evidence: the GPU suite, the driver's bench command (plain and under rocprofv3 --kernel-trace --stats), the counter
# passes of the row kernel, five fresh processes on their first allocation, the secondary kernels
TAG=${1:-r04final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== suite + smoke"
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== bench (driver command)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
cut -c1-1500 $OUT/bench_n1.json
echo "== the same command under rocprofv3 --kernel-trace --stats"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $OUT/bench_n1_under_rocprof.json 2> $OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c 'head -12 {}'
find $OUT/prof -name '*kernel_trace.csv' -delete
echo "== five fresh processes, first allocation"
for k in 1 2 3 4 5; do timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extras --no-e2e --check 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('process $k: K2 %.3f ms frac %.4f  step %.3f ms' % (d['kernel_ms'][r['kernel']], r['frac'], d['ms_per_step']))"; done | tee $OUT/first_allocation_5_processes.txt
echo "== counter passes"
bash scripts/gpu_pmc.sh ${TAG}_pmc "sq1 sq2 sq3 fetch write tcc" 2>&1 | grep -E "k_paf2maf_expand_s\(|k_cigar_stat|rc=" | cut -c1-500
echo "== secondary kernels"
timeout 300 python scripts/gpu_other_kernels.py 100000 5000 2>&1 | grep -E "^K[0-9]" | tee $OUT/other_5k.log
