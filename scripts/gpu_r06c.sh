#!/bin/bash
TAG=${1:-r06c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 120 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "test_maf_pair_stat" -p no:cacheprovider > $OUT/t1.txt 2>&1
rc=$?; echo "rc=$rc"; grep -v "^  File\|^Extension" $OUT/t1.txt | tail -15 | cut -c1-300
[ $rc = 0 ] || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "maf" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests.txt
for g in 0 4 2; do
  echo "== maf_group $g"
  WGA_MAF_GROUP=$g timeout 120 python scripts/gpu_maf_kernels.py 2>&1 | grep -E "^K[34]|blocks"
done | tee $OUT/maf_calls.txt
timeout 200 python scripts/gpu_maf_kernels.py 2000000 1500 2>&1 | grep -E "^K[34]|blocks" | tee -a $OUT/maf_calls.txt
cd /tmp
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $R/scripts/gpu_maf_kernels.py > $OUT/stats.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_maf" in r["Name"][:12]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
