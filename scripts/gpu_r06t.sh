#!/bin/bash
# round 6: instruction / wait counters of K5's kernels at configs[3]'s stated size (the one-call protocol)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export K5_MODE=fused K5_REPS=2
WGA_PMC_CMD="python $R/scripts/gpu_k5_scaling.py 10" bash scripts/gpu_pmc.sh r06t "sq1 sq2" 2>&1 | grep -E "rc=|k_cov" | cut -c1-420
