#!/bin/bash
# round 5, call 12: the list pass with the near record's advance summed by the wave itself and XCD-contiguous tiles
TAG=${1:-r05l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden_py.py -q -m gpu -x -k "(pafcov and not stated_size) or gpu_cov" -p no:cacheprovider 2>&1 | tail -2
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
    for r in rows[-9:]:
        print("%-44s start %12.1f us  dur %10.1f us" % (r["Kernel_Name"][:44], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
