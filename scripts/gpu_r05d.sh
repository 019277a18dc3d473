#!/bin/bash
# round 5, call 4: K1 and K5's list pass as grids of resident waves that request their next tile's ops early; register-budget
# variants of both; the stated-size pafcov job; the headline bench; the whole GPU suite (DevFasta pools padded, reduce-scatter
# by peer reads, K1 / K5 rewritten since the last full run).
TAG=${1:-r05d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== K1"
timeout 300 python scripts/gpu_k1.py k1b4 2>&1 | grep "^K1" | tee $OUT/k1.txt
echo "== K5 at stated size: per-kernel split (product build)"
cd /tmp
K5_MODE=both K5_REPS=2 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/scripts/gpu_k5_scaling.py 10 > $OUT/k5_stated.log 2>&1; echo "rc=$?"
grep -E "^chunks|fused ==" $OUT/k5_stated.log
cd $R
python - <<PY | tee $OUT/k5_kernels.txt
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_cov", "void k_cov", "k_scan", "void k_scan"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows[-19:]:
        print("%-44s start %12.1f us  dur %10.1f us" % (r["Kernel_Name"][:44], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
for v in k5lw4 k5lw6; do
  echo "== variant $v"
  WGA_LIB=$R/build_variants/libwgahip_$v.so K5_MODE=sep K5_REPS=3 timeout 600 python scripts/gpu_k5_scaling.py 10 2>&1 | grep -E "^chunks" | tail -2
done
echo "== bench (headline line)"
timeout 900 python bench.py --no-e2e > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "rc=$?"; cut -c1-1500 $OUT/bench_n1.json
echo "== the whole GPU suite"
timeout 1800 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -5
