#!/bin/bash
# GPU tests (all), then the op-stream kernels on 5 kop records and on 50 kop records (the piece kernels)
TAG=${1:-r03h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $OUT/pytest_gpu.log | tail -12
timeout 600 python scripts/gpu_other_kernels.py 100000 5000 > $OUT/other_5k.log 2>&1; echo "other 5k rc=$?"; grep -E "^K7|^K10|^K12" $OUT/other_5k.log
timeout 600 python scripts/gpu_other_kernels.py 10000 50000 > $OUT/other_50k.log 2>&1; echo "other 50k rc=$?"; grep -E "^K7|^K10|^K12" $OUT/other_50k.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof50k -o p -- python $R/scripts/gpu_other_kernels.py 10000 50000 > $OUT/other_50k_prof.log 2>&1; echo "prof rc=$?"
python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof50k/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if any(k in r["Name"] for k in ("call", "chain", "dotplot", "piece")):
            print("%-60s calls %5s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
