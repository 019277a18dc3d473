#!/bin/bash
# same box: builds of the library, timed with scripts/gpu_other_kernels.py.  Each argument is a WGA_EXTRA_FLAGS string;
# env GREP = which lines to show (default: all)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for F in "$@"; do
  WGA_EXTRA_FLAGS="$F" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1 || { echo "[$F] build failed"; continue; }
  echo "== [$F]"
  timeout 900 python scripts/gpu_other_kernels.py 2>&1 | grep -E "${GREP:-K}"
done
