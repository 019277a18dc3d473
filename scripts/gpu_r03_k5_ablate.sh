#!/bin/bash
# K5 list pass: what it waits for.  Variants without the window-count atomics (1), without publish / look-back (2), without the
# piece stores (4), without the whole piece search (8), without all of them (15) — wrong results, times only.
TAG=${1:-r03B}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for k in 0 1 2 4 8 15; do
  lib=$R/build_variants/libwgahip_k5ab$k.so; [ $k = 0 ] && lib=
  WGA_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$k -o p -- python $R/scripts/gpu_k5_scaling.py 1 > $OUT/ab$k.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/p$k/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_cov_list" in row["Name"]:
            print("ablate $k: k_cov_list_pieces calls %s avg %.1f us" % (row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done 2>&1 | tee $OUT/ablate.log
find $OUT -name '*kernel_trace.csv' -delete
