#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the paf2maf row kernel, v1 against the staged build given in $1 (WGA_EXTRA_FLAGS)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/${2:-k2s_pmc}; mkdir -p $OUT
export TMPDIR=/tmp
WGA_EXTRA_FLAGS="$1" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
cd /tmp
for var in 0 1; do for c in WRITE_SIZE FETCH_SIZE; do
  WGA_EXPAND_VARIANT=$var timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/v${var}_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --check 0 ${BENCH_ARGS} > /dev/null 2> $OUT/v${var}_$c.err; echo "v$var $c rc=$?"
done; done
python - <<PY
import csv, glob, os, collections
for var in (0, 1):
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        for f in glob.glob(os.path.join("$OUT", "v%d_%s" % (var, c), "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "paf2maf_expand" in k: agg[k.split("(")[0]].append(float(row["Counter_Value"]))
            for k, v in agg.items():
                print("variant %d %-11s %-28s launches %d  per launch %.4g KB" % (var, c, k, len(v), sum(v) / len(v)))
PY
find $OUT -name '*kernel_trace.csv' -delete
