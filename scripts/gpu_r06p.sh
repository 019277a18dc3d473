#!/bin/bash
# round 6: the driver's bench command and the same command under rocprofv3 (allocation spread with as many launches per allocation as the headline)
TAG=${1:-r06p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_n1_under_rocprof.json 2> $OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c 'head -3 {}' | cut -c1-160
find $OUT/prof -name '*kernel_trace.csv' -delete
python - <<PY
import json
for f in ("bench_n1.json", "bench_n1_under_rocprof.json"):
    r = json.load(open("$OUT/" + f)); sp = r["roofline"]["allocation_spread"]
    print(f, "value %.4g" % r["value"], "ms/step %.3f" % r["ms_per_step"], [round(x, 3) for x in sp["k_ms_by_allocation"]], "median %.3f" % sp["k_ms_median"], "mean %.3f" % (sum(sp["k_ms_by_allocation"]) / 5))
PY
