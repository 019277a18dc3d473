#!/bin/bash
# round 6: K5 — tile info by records, the full-window walk with v_min; tiles of 2 048 ops as A/B builds; kernels one by one
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06x
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pafcov and not stated" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
export TMPDIR=/tmp K5_MODE=fused K5_REPS=3
for v in product k5t11 k5t11f; do
  echo "== $v"
  if [ $v = product ]; then unset WGA_LIB; else export WGA_LIB=$R/build_variants/libwgahip_$v.so; fi
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o p -- python $R/scripts/gpu_k5_scaling.py 10 > $OUT/run_$v.txt 2>&1 )
  grep accumulate_final $OUT/run_$v.txt | tail -2
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cov" in r["Name"]: print(r["Name"][:44], r["Calls"], "avg %.1f us" % (float(r["AverageNs"])/1e3))
PY
  find $OUT/prof_$v -name '*kernel_trace.csv' -delete
done
