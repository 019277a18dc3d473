#!/bin/bash
# same box: bench line + secondary-kernel timings for several builds (each argument = extra compiler flags)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for F in "$@"; do
  echo "=== build [$F]"
  WGA_EXTRA_FLAGS="$F" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --no-extras --check 0 --steps 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench K1 %.3f ms  K2 %.3f ms (frac %.3f)  step %.3f ms  value %.3e' % (d['kernel_ms']['k_cigar_stat'], d['kernel_ms']['k_paf2maf_expand'], d['roofline']['frac'], d['ms_per_step'], d['value']))"
  done
  python scripts/gpu_maf_kernels.py 2000000 1500 2>/dev/null | grep "^K"
  python scripts/gpu_other_kernels.py 2>/dev/null | grep "^K"
done
