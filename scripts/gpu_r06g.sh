#!/bin/bash
# round 6: K5 at configs[3]'s stated size (LDS window padded; what the cold first call is made of); `call` on MAF at 2 M blocks with the
# copy-out on a helper thread; the GPU suite on this tree
TAG=${1:-r06g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
K5_COLD=1 timeout 600 python scripts/gpu_k5_scaling.py 10 2>&1 | grep -v "^W2\|^E2" | tee $OUT/k5_stated.log
timeout 900 python scripts/gpu_e2e_at_size.py maf-only 2>&1 | tail -9 | tee $OUT/e2e.txt
timeout 3000 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/gpu_suite.txt
