#!/bin/bash
# round 5, call 6: instruction and wait counters of K1 and of K5's list pass (what bounds two kernels that read 4 B per op at
# half the rate a plain read reaches)
TAG=${1:-r05f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
WGA_PMC_CMD="python $R/scripts/gpu_k1.py" timeout 600 bash scripts/gpu_pmc.sh ${TAG}_k1 "sq1 sq2 sq3" 2>&1 | grep -E "k_cigar_stat|k_tile_rec|rc=" | cut -c1-420 | tee $OUT/k1_pmc.log
WGA_PMC_CMD="env K5_MODE=sep K5_REPS=2 python $R/scripts/gpu_k5_scaling.py 1" timeout 900 bash scripts/gpu_pmc.sh ${TAG}_k5 "sq1 sq2 sq3" 2>&1 | grep -E "k_cov_list|k_cov_windows|rc=" | cut -c1-420 | tee $OUT/k5_pmc.log
