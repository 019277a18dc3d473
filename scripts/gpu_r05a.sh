#!/bin/bash
# round 5, call 1: K5 after the instruction diet (32-bit replay walk, one scan per tile in the list pass) and with the
# marks -> counts scan inside the replay: GPU parity of the small cases, the per-kernel split and the HBM counter passes of
# configs[3] at its stated size (10 chunks), then the whole GPU suite.
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== K5 parity (small cases)"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "pafcov and not stated_size" -p no:cacheprovider 2>&1 | tail -3
echo "== K5 at stated size: per-kernel split"
cd /tmp
K5_MODE=both K5_REPS=2 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/scripts/gpu_k5_scaling.py 10 > $OUT/k5_stated.log 2>&1; echo "rc=$?"
cat $OUT/k5_stated.log | grep -v "^$" | tail -12
cd $R
python - <<PY | tee $OUT/k5_kernels.txt
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_cov", "void k_cov", "k_scan", "void k_scan"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        print("%-44s start %12.1f us  dur %10.1f us" % (r["Kernel_Name"][:44], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $OUT -name '*kernel_trace.csv' -size +5M -delete
echo "== K5 at stated size: HBM counters of the fused call (2 launches: the first sizes the work lists)"
WGA_PMC_CMD="env K5_MODE=fused K5_REPS=2 python $R/scripts/gpu_k5_scaling.py 10" timeout 1500 bash scripts/gpu_pmc.sh ${TAG}_k5pmc "fetch write sq1" 2>&1 | grep -E "k_cov|rc=" | cut -c1-400 | tee $OUT/k5_pmc.log
echo "== the whole GPU suite"
timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -5
