#!/bin/bash
# the bench line with the job-placed arena (twice: two processes, two sets of candidates), GPU tests of the new entry points
TAG=${1:-r03l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "arena or placed or drain" > $OUT/pytest_place.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_place.log
for rep in 1 2; do
timeout 1500 python bench.py --no-e2e > $OUT/bench$rep.json 2> $OUT/bench$rep.err; echo "bench rc=$?"; python - <<PY
import json
r = json.loads([l for l in open("$OUT/bench$rep.json").read().splitlines() if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["kernel_ms"], r["roofline"]["frac"], r["roofline"].get("frac_first_allocation"), r["output_placement"])
PY
done
