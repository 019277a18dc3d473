#!/bin/bash
# round 5, call 7: K5 with tiles of 2 048 ops (32 per lane) against 1 024, and two register budgets of the list pass
TAG=${1:-r05g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== K5 parity (small cases, golden fixtures)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_py.py -q -m gpu -x -k "(pafcov and not stated_size) or gpu_cov" -p no:cacheprovider 2>&1 | tail -3
echo "== K5 at stated size: per-kernel split (product build: 2 048-op tiles)"
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
for v in k5t1024 k5t2048w5 k5t2048w3; do
  echo "== variant $v"
  WGA_LIB=$R/build_variants/libwgahip_$v.so K5_MODE=both K5_REPS=3 timeout 600 python scripts/gpu_k5_scaling.py 10 2>&1 | grep -E "^chunks" | tail -2
done
