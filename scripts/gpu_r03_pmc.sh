#!/bin/bash
# counters of the row kernel for one expand_variant: usage gpurun -- 'bash scripts/gpu_r03_pmc.sh TAG VARIANT'
TAG=${1:-r03pmc}; V=${2:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export WGA_EXPAND_VARIANT=$V
export WGA_EXPAND_DRAIN_MIN=64
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-placement-probe --check 0"
run() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $BENCH > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run sq3 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in ("sq1","sq2","sq3","fetch","write"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        seen = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:32]
            if not k.startswith("k_paf2maf"): continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            seen[k].add(row.get("Dispatch_Id"))
        for k, v in agg.items():
            n = max(1, len(seen[k]))
            print(d, k, "launches", n, {c: "%.4g" % (x / n) for c, x in v.items()})
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
