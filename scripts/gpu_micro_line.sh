#!/bin/bash
# line-split / LDS-staging copy micro-benchmark + its WRITE_SIZE / FETCH_SIZE per variant
TAG=${1:-mline}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R/scripts/micro
[ -x line_split_copy ] || hipcc --offload-arch=gfx950 -O3 -o line_split_copy line_split_copy.hip
timeout 300 ./line_split_copy > $OUT/times.txt 2>&1; echo "micro rc=$?"; cat $OUT/times.txt
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- $R/scripts/micro/line_split_copy > /dev/null 2> $OUT/$c.err; echo "$c rc=$?"
done
python - <<PY
import csv, glob, os, collections
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    for f in glob.glob(os.path.join("$OUT", c, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.OrderedDict()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            agg.setdefault(k, []).append(float(row["Counter_Value"]))
        for k, v in agg.items():
            # 6 launches per (variant, shift config); 4 shift configs in order
            per = [sum(v[i:i+6]) / 6 for i in range(0, len(v), 6)]
            print(c, k[:60], ["%.4g" % x for x in per])
PY
find $OUT -name '*kernel_trace.csv' -delete
