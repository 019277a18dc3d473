#!/bin/bash
# one counter pass of the row kernel: usage gpurun -- 'bash scripts/gpu_r03_pmc2.sh TAG VARIANT [records mean pool]'
TAG=${1:-r03pmc2}; V=${2:-2}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export WGA_EXPAND_VARIANT=$V
export WGA_EXPAND_DRAIN_MIN=64
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/c -o p -- python $R/scripts/gpu_k2_one.py $R/wgatools_amd/libwgahip.so "$@" > $OUT/run.log 2> $OUT/run.err
tail -1 $OUT/run.log
cd $R
python - <<PY
import csv, glob, collections, os
for f in glob.glob(os.path.join("$OUT", "c", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:28]
        if not k.startswith("k_paf2maf_expand"): continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); seen[k].add(row.get("Dispatch_Id"))
    for k, v in agg.items():
        n = max(1, len(seen[k]))
        print(k, "launches", n, {c: "%.4g" % (x / n) for c, x in v.items()})
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
