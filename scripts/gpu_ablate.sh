#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-abl}
mkdir -p $OUT; cd $R
# needs a build with WGA_EXTRA_FLAGS=-DWGA_PROFILE.  1 = phase A only, 8 = no row jobs, 16 = no complex path,
# 2 = no loads, 4 = no stores, 32 = no fast path (everything through the queue)
for cfg in "" "--param expand_ablate=1" "--param expand_ablate=8" "--param expand_ablate=16" "--param expand_ablate=2" "--param expand_ablate=4" "--param expand_ablate=6" "--param expand_ablate=22"; do
  timeout 300 python bench.py --no-cpu-baseline --check 0 --steps 3 --warmup 1 $cfg > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<PY
import json,sys
try:
    d = json.load(open("$OUT/b.json")); print("%-28s expand %.2f ms  stat %.2f ms" % (sys.argv[1], d["kernel_ms"]["k_paf2maf_expand"], d["kernel_ms"]["k_cigar_stat"]))
except Exception as e: print(sys.argv[1], "failed", e, open("$OUT/b.err").read()[-500:])
PY
done
