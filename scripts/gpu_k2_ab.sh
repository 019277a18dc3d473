#!/bin/bash
# same box: builds of the paf2maf row kernel.  Each argument is "<variant>|<WGA_EXTRA_FLAGS>" (variant 0 = v1, 2 = the window kernel)
# env: CFGS (bench configurations, ';'-separated), REPS
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
IFS=';' read -ra CF <<< "${CFGS:-;--records 10000 --mean-ops 50000;--records 1000000 --mean-ops 500;--pool-mb 1000}"
REPS=${REPS:-1}
for A in "$@"; do
  V=${A%%|*}; F=${A#*|}
  WGA_EXTRA_FLAGS="$F" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1 || { echo "[$A] build failed"; continue; }
  for cfg in "${CF[@]}"; do for rep in $(seq $REPS); do
    WGA_EXPAND_VARIANT=$V python bench.py --no-cpu-baseline --no-extras --check 0 --steps 8 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[v$V $F] [$cfg] K2 %.3f ms frac %.3f  step %.3f ms' % (d['kernel_ms']['k_paf2maf_expand'], d['roofline']['frac'], d['ms_per_step']))"
  done; done
done
