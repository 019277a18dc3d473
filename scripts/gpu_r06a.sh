#!/bin/bash
# round 6, call 1: where the MAF walks stand — kernel-only durations (rocprofv3) next to the call times, and a counter pass
TAG=${1:-r06a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 300 python scripts/gpu_maf_kernels.py 2>&1 | grep -E "^K[34]|blocks" | tee $OUT/maf_calls.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $R/scripts/gpu_maf_kernels.py > $OUT/stats.log 2>&1
grep -h "k_maf\|k_scan\|Name" $(find $OUT/stats -name '*kernel_stats.csv') | cut -c1-60,200- | head -20
for name in sq1 sq2; do
  if [ $name = sq1 ]; then C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"; else C="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; fi
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$name -o p -- python $R/scripts/gpu_maf_kernels.py > $OUT/$name.log 2>&1
  echo "$name rc=$?"
done
cd $R
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in ("sq1","sq2"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:24]
            if not k.startswith("k_maf"): continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k,row["Counter_Name"])] += 1
        for k, v in agg.items():
            print(d, k, {c: "%.4g/%d" % (x, n[(k,c)]) for c, x in v.items()})
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
find $OUT -name '*counter_collection.csv' -size +5M -delete
