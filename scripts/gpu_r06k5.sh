#!/bin/bash
# round 6: K5 after its second rebuild (16 K windows, 16-byte descriptors, marks through dummy words, tile info by records):
# every pafcov test incl. the stated size, the call's time, its kernels one by one, the traffic and instruction counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06k5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -x -q -m gpu -s -k "pafcov" > $OUT/tests.txt 2>&1
grep -E "config 4 at size|passed|failed" $OUT/tests.txt | tail -4
K5_MODE=both K5_REPS=3 timeout 400 python scripts/gpu_k5_scaling.py 10 2>&1 | grep -E "accumulate|fused ==|hipMalloc" | tee $OUT/stated.log
timeout 300 python scripts/gpu_other_kernels.py 2>&1 | grep "^K5" | tee $OUT/small.txt
export K5_MODE=fused K5_REPS=2
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/scripts/gpu_k5_scaling.py 10 > $OUT/run.txt 2>&1 )
python - <<PY | tee $OUT/kernels.txt
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cov" in r["Name"] or "k_scan" in r["Name"]: print(r["Name"][:48], r["Calls"], "avg %.1f us  min %.1f  max %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
find $OUT/prof -name '*kernel_trace.csv' -delete
WGA_PMC_CMD="python $R/scripts/gpu_k5_scaling.py 10" bash scripts/gpu_pmc.sh r06k5_pmc "sq1 sq2 fetch write" 2>&1 | grep -E "rc=|k_cov" | cut -c1-420 | tee $OUT/pmc.txt
