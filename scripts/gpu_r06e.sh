#!/bin/bash
# round 6: `call` on MAF with the rules and rows on the device (K19): the CLI's call tests on the GPU, then the command at configs[2]'s size
TAG=${1:-r06e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_cli.py -q -m gpu -x -k "call or maf" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests.txt
timeout 900 python scripts/gpu_e2e_at_size.py maf-only 2>&1 | tail -12 | tee $OUT/e2e.txt
