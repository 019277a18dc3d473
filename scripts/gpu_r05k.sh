#!/bin/bash
# round 5, call 11: where the list pass's 28 ms go — A/B builds (1: sum published by a store; the others give wrong results and
# only time: 2 no window-count atomics, 4 no piece store, 8 no look-back, 14 all three), the list pass's duration from rocprofv3
TAG=${1:-r05k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
for v in product k5abl1 k5abl2 k5abl4 k5abl8 k5abl14; do
  if [ "$v" != product ]; then export WGA_LIB=$R/build_variants/libwgahip_$v.so; else unset WGA_LIB; fi
  K5_MODE=sep K5_REPS=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o p -- python $R/scripts/gpu_k5_scaling.py 10 > $OUT/$v.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("k_cov_list_pieces"):
            print("%-10s list pass: %s calls, min %.2f ms, avg %.2f ms" % ("$v", r["Calls"], float(r["MinNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
  find $OUT/prof_$v -name '*kernel_trace.csv' -delete
done 2>&1 | tee $OUT/ablations.txt
