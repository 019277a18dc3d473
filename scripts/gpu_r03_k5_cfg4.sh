#!/bin/bash
# per-kernel split of ONE accumulate call at configs[3]'s stated size (the test the driver's suite runs)
TAG=${1:-r03D}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python -m pytest $R/tests/test_gpu_parity.py -q -m gpu -s -k "config4_at_stated_size" -p no:cacheprovider > $OUT/cfg4.log 2>&1; echo "rc=$?"
grep -E "config 4|passed|failed" $OUT/cfg4.log
cd $R
python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_cov", "k_scan", "void k_scan"))]
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows:
        print("%-28s start %10.1f us  dur %10.1f us" % (r["Kernel_Name"][:28], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
