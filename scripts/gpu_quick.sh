#!/bin/bash
# quick loop: parity subset + bench (no profiler)
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "${2:-paf2maf or stat}" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
try:
    d = json.load(open("$OUT/bench.json"))
    print("value %.3e ops/s  ms/step %.2f  kernels %s  frac %.3f  stat GB/s %.0f" % (d["value"], d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"], d["roofline"]["k_cigar_stat_GBps"]))
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/bench.err").read()[-2000:])
PY
