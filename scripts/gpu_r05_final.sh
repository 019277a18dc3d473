#!/bin/bash
# round 5 evidence: the GPU suite (with configs[3] at its stated size through both pafcov protocols), smoke, the driver's bench
# command (plain and under rocprofv3 --kernel-trace --stats), the counter passes of the row kernel and of K5 at the stated
# size, the secondary kernels, the command line file to file at size
TAG=${1:-r05final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== suite + smoke"
timeout 1800 python -m pytest tests -x -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "config 4 at size|passed|failed" $OUT/pytest_gpu.log | tail -5
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== bench (driver command)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
cut -c1-1200 $OUT/bench_n1.json
echo "== the same shape under rocprofv3 --kernel-trace --stats (no other shapes in the process: the row kernel's row holds warm-up + timed launches of ONE shape)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_n1_under_rocprof.json 2> $OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c 'head -8 {}' | cut -c1-200
find $OUT/prof -name '*kernel_trace.csv' -delete
echo "== counter passes: the row kernel"
bash scripts/gpu_pmc.sh ${TAG}_pmc "sq1 sq2 fetch write" 2>&1 | grep -E "k_paf2maf_expand_s\(|k_cigar_stat|rc=" | cut -c1-400
echo "== counter passes: K5 at the stated size (fused call, 2 launches: the first sizes the work lists)"
WGA_PMC_CMD="env K5_MODE=fused K5_REPS=2 python $R/scripts/gpu_k5_scaling.py 10" timeout 1500 bash scripts/gpu_pmc.sh ${TAG}_k5pmc "fetch write" 2>&1 | grep -E "k_cov|rc=" | cut -c1-300 | tee $OUT/k5_pmc.log
echo "== secondary kernels"
timeout 300 python scripts/gpu_other_kernels.py 100000 5000 2>&1 | grep -E "^K[0-9]" | tee $OUT/other_5k.log
timeout 300 python scripts/gpu_maf_kernels.py 2>&1 | tail -12 | tee $OUT/maf_kernels.log
echo "== file to file at size"
timeout 900 python scripts/gpu_e2e_at_size.py > $OUT/e2e_at_size.log 2>&1; echo "e2e rc=$?"; tail -30 $OUT/e2e_at_size.log | cut -c1-300
