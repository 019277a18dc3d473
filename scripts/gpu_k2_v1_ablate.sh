#!/bin/bash
# VALU / SALU wave-instructions and time of v1's parts: a -DWGA_PROFILE build with expand_ablate = 1 (phase A only), 8 (+ segment
# set-up, no row jobs), 16 (no complex path), 0 (everything)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/${1:-k2v1_abl}; mkdir -p $OUT
export TMPDIR=/tmp
WGA_EXTRA_FLAGS="-DWGA_PROFILE" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
for abl in 1 8 16 0; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a$abl -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --check 0 --param expand_ablate=$abl > /dev/null 2> $OUT/a$abl.err)
  python bench.py --no-cpu-baseline --no-extras --check 0 --steps 8 --param expand_ablate=$abl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ablate $abl: K2 %.3f ms' % d['kernel_ms']['k_paf2maf_expand'])"
  python - <<PY
import csv, glob, os, collections
for f in glob.glob(os.path.join("$OUT", "a$abl", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("k_paf2maf_expand("): agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("ablate $abl:", {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(agg.items())})
PY
done
find $OUT -name '*kernel_trace.csv' -delete
python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
