#!/bin/bash
# round 6 evidence: the GPU suite, smoke, the driver's bench command (plain and under rocprofv3 --kernel-trace --stats), the
# counter passes of the row kernel (FETCH / WRITE: roofline.traffic), the MAF walks, the secondary kernels, the command line at size
TAG=${1:-r06final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== suite + smoke"
timeout 2400 python -m pytest tests -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "1e8-column|reduce_scatter_i32 over|config 4 at size|passed|failed" $OUT/pytest_gpu.log | tail -8
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== bench (driver command)"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
cut -c1-600 $OUT/bench_n1.json
echo "== the same command's kernels under rocprofv3 --kernel-trace --stats (one shape in the process: the row kernel's row = warm-up + timed launches + the allocation spread)"
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_n1_under_rocprof.json 2> $OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c 'head -8 {}' | cut -c1-200
find $OUT/prof -name '*kernel_trace.csv' -delete
echo "== counter passes: the row kernel"
bash scripts/gpu_pmc.sh ${TAG}_pmc "sq1 sq2 fetch write" 2>&1 | grep -E "k_paf2maf_expand_s\(|k_cigar_stat|rc=" | cut -c1-400
echo "== MAF walks"
timeout 300 python scripts/gpu_maf_kernels.py 2>&1 | grep -E "^K[34]|blocks" | tee $OUT/maf_kernels.log
timeout 300 python scripts/gpu_maf_kernels.py 2000000 1500 2>&1 | grep -E "^K[34]|blocks" | tee -a $OUT/maf_kernels.log
timeout 300 python scripts/gpu_maf_kernels.py 20000 15000 2>&1 | grep -E "^K[34]|blocks" | tee -a $OUT/maf_kernels.log
timeout 200 python scripts/gpu_maf_long_block.py 2>&1 | grep -E "^K[34]|one block" | tee -a $OUT/maf_kernels.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_maf -o p -- python $R/scripts/gpu_maf_kernels.py > /dev/null 2>&1 )
python - <<PY | tee -a $OUT/maf_kernels.log
import csv, glob
print("# rocprofv3 --kernel-trace --stats, scripts/gpu_maf_kernels.py (200 000 x 1 500): name calls avg_ns min_ns max_ns")
for f in glob.glob("$OUT/prof_maf/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_maf" in r["Name"][:12]: print(r["Name"][:48], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
find $OUT/prof_maf -name '*kernel_trace.csv' -delete
echo "== secondary kernels"
timeout 300 python scripts/gpu_other_kernels.py 100000 5000 2>&1 | grep -E "^K[0-9]" | tee $OUT/other_5k.log
echo "== file to file at size"
timeout 1200 python scripts/gpu_e2e_at_size.py > $OUT/e2e_at_size.log 2>&1; echo "e2e rc=$?"; tail -30 $OUT/e2e_at_size.log | cut -c1-300
