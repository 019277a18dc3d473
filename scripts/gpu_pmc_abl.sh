#!/bin/bash
# instruction counts of k_paf2maf_expand per ablation level
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmcabl}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for abl in 0 1 8 16 32; do
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a$abl -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --check 0 --param expand_ablate=$abl > $OUT/a$abl.json 2> $OUT/a$abl.err
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$OUT/a$abl/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("k_paf2maf_expand"):
            agg[row["Counter_Name"]] += float(row["Counter_Value"]) / 2
print("ablate=$abl", {k: "%.3g" % v for k, v in sorted(agg.items())})
PY
  rm -rf $OUT/a$abl
done
