#!/bin/bash
# round 5, call 13: the list pass with the tile sums of a block handed over in LDS (4 / 8 / 16 waves per block, with and without
# XCD-contiguous tiles); the list pass's duration from rocprofv3 --stats
TAG=${1:-r05m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden_py.py -q -m gpu -x -k "(pafcov and not stated_size) or gpu_cov" -p no:cacheprovider 2>&1 | tail -2
cd /tmp
for v in product k5bw8 k5bw4x k5bw8x k5bw16; do
  if [ "$v" != product ]; then export WGA_LIB=$R/build_variants/libwgahip_$v.so; else unset WGA_LIB; fi
  K5_MODE=both K5_REPS=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o p -- python $R/scripts/gpu_k5_scaling.py 10 > $OUT/$v.log 2>&1
  grep -E "^chunks.*rep 1|fused ==" $OUT/$v.log | cut -c1-150
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("k_cov_list_pieces"):
            print("%-10s list pass: %s calls, min %.2f ms, avg %.2f ms" % ("$v", r["Calls"], float(r["MinNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
  find $OUT/prof_$v -name '*kernel_trace.csv' -delete
done 2>&1 | tee $OUT/variants.txt
