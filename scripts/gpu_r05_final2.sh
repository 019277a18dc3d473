#!/bin/bash
# round 5, final evidence for K5 after the LDS hand-over of the list pass: the GPU suite, the stated-size job per kernel, the
# HBM counter passes of the one-call protocol
TAG=${1:-r05final2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "config 4 at size|passed|failed" $OUT/pytest_gpu.log | tail -5
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
cd /tmp
K5_MODE=both K5_REPS=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/scripts/gpu_k5_scaling.py 10 > $OUT/k5_stated.log 2>&1; echo "rc=$?"
grep -E "^chunks|fused ==" $OUT/k5_stated.log
cd $R
python - <<PY | tee $OUT/k5_kernels.txt
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_cov", "void k_cov", "k_scan", "void k_scan"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows[-19:]:
        print("%-44s start %12.1f us  dur %10.1f us" % (r["Kernel_Name"][:44], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
WGA_PMC_CMD="env K5_MODE=fused K5_REPS=2 python $R/scripts/gpu_k5_scaling.py 10" timeout 900 bash scripts/gpu_pmc.sh ${TAG}_k5pmc "fetch write" 2>&1 | grep -E "k_cov|rc=" | cut -c1-300 | tee $OUT/k5_pmc.log
timeout 300 python scripts/gpu_other_kernels.py 100000 5000 2>&1 | grep -E "^K5" | tee $OUT/other_k5.log
