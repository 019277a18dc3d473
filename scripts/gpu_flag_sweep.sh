#!/bin/bash
# same box, several builds of libwgahip.so: each argument is one set of extra compiler flags.
# usage: gpu_flag_sweep.sh "-DA=0 -DB=0" "-DA=1 -DB=0" ...   (K2 time / roofline fraction on three record shapes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for F in "$@"; do
  WGA_EXTRA_FLAGS="$F" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
  for cfg in "" "--records 10000 --mean-ops 50000" "--records 1000000 --mean-ops 500"; do
    for rep in 1 2; do
      python bench.py --no-cpu-baseline --no-extras --check 0 --steps 8 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$F] [$cfg] K2 %.3f ms frac %.3f  step %.3f ms' % (d['kernel_ms']['k_paf2maf_expand'], d['roofline']['frac'], d['ms_per_step']))"
    done
  done
done
