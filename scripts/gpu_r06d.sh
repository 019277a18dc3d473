#!/bin/bash
TAG=${1:-r06d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "maf_block_of_1e8 or maf_long_blocks" -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension" | tail -8 | tee $OUT/tests.txt
timeout 120 python scripts/gpu_maf_long_block.py 2>&1 | tail -8 | tee $OUT/long_block.txt
timeout 120 python scripts/gpu_maf_kernels.py 2>&1 | grep -E "^K[34]|blocks" | tee $OUT/maf_calls.txt
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $R/scripts/gpu_maf_long_block.py > $OUT/stats.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_maf" in r["Name"][:12]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
