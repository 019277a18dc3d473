#!/bin/bash
# round 6: K5's kernels at the stated size, one by one (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06w
mkdir -p $OUT
export TMPDIR=/tmp K5_MODE=fused K5_REPS=3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/scripts/gpu_k5_scaling.py 10 > $OUT/run.txt 2>&1 )
grep accumulate_final $OUT/run.txt
python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cov" in r["Name"] or "k_scan" in r["Name"]: print(r["Name"][:44], r["Calls"], "avg %.1f us  min %.1f  max %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
find $OUT/prof -name '*kernel_trace.csv' -delete
