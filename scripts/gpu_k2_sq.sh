#!/bin/bash
# SQ counter passes of the paf2maf row kernel, v1 against the staged build given in $1 (WGA_EXTRA_FLAGS)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/${2:-k2_sq}; mkdir -p $OUT
export TMPDIR=/tmp
WGA_EXTRA_FLAGS="$1" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
cd /tmp
run() { # var name counters...
  var=$1; name=$2; shift; shift
  WGA_EXPAND_VARIANT=$var timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/v${var}_$name -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --check 0 ${BENCH_ARGS} > /dev/null 2> $OUT/v${var}_$name.err; echo "v$var $name rc=$?"
}
for var in 0 1; do
  run $var sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
  run $var sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
  run $var sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC
done
python - <<PY
import csv, glob, os, collections
for var in (0, 1):
    for name in ("sq1", "sq2", "sq3"):
        for f in glob.glob(os.path.join("$OUT", "v%d_%s" % (var, name), "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0]
                if k in ("k_paf2maf_expand", "k_paf2maf_expand_p"): agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, v in agg.items():
                for c, x in sorted(v.items()):
                    if sum(x): print("variant %d %-20s %-24s %.4g per launch" % (var, k, c, sum(x) / len(x)))
PY
find $OUT -name '*kernel_trace.csv' -delete
