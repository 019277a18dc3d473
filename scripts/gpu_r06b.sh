#!/bin/bash
# round 6: the MAF stream walks (K3 / K4 rewritten): parity tests that touch them, call times, kernel-only durations
TAG=${1:-r06b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "maf" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests.txt
for g in 0 4 2; do
  echo "== maf_group $g"
  WGA_MAF_GROUP=$g timeout 300 python scripts/gpu_maf_kernels.py 2>&1 | grep -E "^K[34]|blocks"
done | tee $OUT/maf_calls.txt
timeout 300 python scripts/gpu_maf_kernels.py 2000000 1500 2>&1 | grep -E "^K[34]|blocks" | tee -a $OUT/maf_calls.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $R/scripts/gpu_maf_kernels.py > $OUT/stats.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(("void k_maf", "k_maf")): print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
