#!/bin/bash
# the headline bench on variants of configs[1] (strand mix, M-only ops, record length)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-var}; mkdir -p $OUT; cd $R
for cfg in "" "--neg-frac 0" "--neg-frac 1" "--m-only" "--records 10000 --mean-ops 50000" "--records 1000000 --mean-ops 500" "--records 2000 --mean-ops 250000"; do
  timeout 400 python bench.py --no-cpu-baseline --check 4 $cfg > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<PY
import json,sys
try:
    d = json.load(open("$OUT/b.json")); print("%-36s %.3e ops/s  step %.2f ms  K2 %.2f ms  frac %.3f  K1 %.2f ms  (%.2e ops)" % (sys.argv[1] or "(default)", d["value"], d["ms_per_step"], d["kernel_ms"]["k_paf2maf_expand"], d["roofline"]["frac"], d["kernel_ms"]["k_cigar_stat"], d["config"]["ops_per_gpu"]))
except Exception as e: print(sys.argv[1], "failed", e, open("$OUT/b.err").read()[-400:])
PY
done
