#!/bin/bash
# PMC passes for the bench kernels (own runs, --kernel-trace only; see MI355X_MICROARCH.md §HBM).
TAG=${1:-pmc01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-e2e --no-placement-probe --check 0"
# another command's kernels (e.g. K5: WGA_PMC_CMD="python $R/scripts/gpu_k5_scaling.py 1" bash scripts/gpu_pmc.sh k5pmc "sq1 sq2 sq3 tcc")
BENCH=${WGA_PMC_CMD:-$BENCH}
PASSES=${2:-"sq1 sq2 fetch write tcc"}
want() { [[ " $PASSES " == *" $1 "* ]]; }
run() { # name counters...
  name=$1; shift
  want $name || return 0
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $BENCH > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq3 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $R
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in ("sq1","sq2","sq3","fetch","write","tcc"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            if not (k.startswith("k_") or "k_" in k[:8]): continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        for k, v in agg.items():
            print(d, k, {c: "%.4g" % x for c, x in v.items()})
PY
# keep only the small per-kernel counter files
find $OUT -name '*kernel_trace.csv' -size +5M -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete   # (gpurun copies back at most 64 MiB)
