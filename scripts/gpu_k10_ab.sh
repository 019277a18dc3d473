#!/bin/bash
# same box: builds of the paf2chain kernel.  Each argument is a WGA_EXTRA_FLAGS string ("" = the tree's defaults).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for F in "$@"; do
  WGA_EXTRA_FLAGS="$F" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1 || { echo "[$F] build failed"; continue; }
  echo "== [$F]"
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "cigar_chain or runs_bridge or maf2chain" 2>&1 | tail -2
  timeout 600 python scripts/gpu_k10.py 2>&1 | grep K10
done
