#!/bin/bash
# What the next round should run first (one gpurun call, ~9 GPU-minutes): the suite on the tree as it stands, the secondary
# kernels' rates (K5 / K6 / K7 / K10 / K11 / K12 after round 3's late changes), K5's kernel timeline at configs[3]'s size, and
# the counter pass on K5's list pass that round 3 had no minutes left for.
#   gpurun --timeout 900 -- 'bash scripts/gpu_next_round_first_call.sh r04a'
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== suite + smoke"
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== secondary kernels, 5 kop records"
timeout 300 python scripts/gpu_other_kernels.py 100000 5000 2>&1 | grep -E "^K[0-9]" | tee $OUT/other_5k.log
echo "== K5: scaling and the kernels of one call at configs[3]'s size"
timeout 300 python scripts/gpu_k5_scaling.py 1 2 4 2>&1 | grep -v amdgpu | tee $OUT/k5_scaling.log
bash scripts/gpu_r03_k5_cfg4.sh ${TAG}_cfg4 2>&1 | tail -14
echo "== K5 list pass: counters (SQ mix, waits, TCC requests) over one 10 GB call"
WGA_PMC_CMD="python $R/scripts/gpu_k5_scaling.py 1" bash scripts/gpu_pmc.sh ${TAG}_k5pmc "sq1 sq2 sq3 tcc" 2>&1 | grep -E "k_cov|rc=" | cut -c1-400 | tee $OUT/k5_pmc.log
