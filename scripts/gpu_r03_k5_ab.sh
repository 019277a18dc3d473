#!/bin/bash
# K5 list pass: register allocation for 7 waves per SIMD (65 VGPRs) against 8 (62 VGPRs + 12 bytes of scratch)
TAG=${1:-r03z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cov" -s > $OUT/pytest_cov.log 2>&1; echo "pytest rc=$?"; grep -E "config|passed|failed" $OUT/pytest_cov.log | tail -6
timeout 600 python scripts/gpu_other_kernels.py 100000 5000 2>&1 | grep "^K5" | tee $OUT/k5_5k.log
for lib in "" build_variants/libwgahip_k5w8.so; do
  echo "lib=${lib:-product}" | tee -a $OUT/k5_ab.log
  WGA_LIB=$lib timeout 600 python scripts/gpu_k5_scaling.py 1 4 2>&1 | grep -v amdgpu | tee -a $OUT/k5_ab.log
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/scripts/gpu_k5_scaling.py 2 > $OUT/k5_prof.log 2>&1; echo "prof rc=$?"
cd $R
python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cov" in row["Name"]:
            print("%-50s calls %4s avg %10.1f us" % (row["Name"][:50], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
