#!/bin/bash
# instruction counts and times of the window kernel's parts (ablation builds: wrong output, measurements only)
TAG=${1:-r03abl}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export WGA_EXPAND_VARIANT=2
for v in wab1 wab2 wab3 tree; do
  lib=$R/build_variants/libwgahip_$v.so; [ $v = tree ] && lib=$R/wgatools_amd/libwgahip.so
  cp $R/wgatools_amd/libwgahip.so /tmp/libwgahip_keep.so 2>/dev/null
  WGA_LIB=$lib timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/$v -o p -- python $R/scripts/gpu_k2_one.py $lib > $OUT/$v.log 2> $OUT/$v.err
  echo "$v rc=$?"; tail -2 $OUT/$v.log
done
cd $R
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in ("wab1","wab2","wab3","tree"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:28]
            if not k.startswith("k_paf2maf_expand_w"): continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); seen[k].add(row.get("Dispatch_Id"))
        for k, v in agg.items():
            n = max(1, len(seen[k]))
            print(d, k, "launches", n, {c: "%.4g" % (x / n) for c, x in v.items()})
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
