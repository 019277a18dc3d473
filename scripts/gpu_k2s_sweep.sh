#!/bin/bash
# same box: v1 of the paf2maf row kernel against builds of the staged one (each argument = one set of WGA_EXTRA_FLAGS)
# usage: gpu_k2s_sweep.sh "" "-DWGA_K2S_BLOCKS=6" ...      env: CFGS (bench configurations, ';'-separated), REPS
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
IFS=';' read -ra CF <<< "${CFGS:-;--records 10000 --mean-ops 50000;--records 1000000 --mean-ops 500;--pool-mb 1000}"
REPS=${REPS:-2}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$1] [$2] K2 %.3f ms frac %.3f  step %.3f ms' % (d['kernel_ms']['k_paf2maf_expand'], d['roofline']['frac'], d['ms_per_step']))"; }
python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
for cfg in "${CF[@]}"; do for rep in $(seq $REPS); do
  WGA_EXPAND_VARIANT=0 python bench.py --no-cpu-baseline --check 0 --steps 8 $cfg 2>/dev/null | line "v1" "$cfg"
done; done
for F in "$@"; do
  WGA_EXTRA_FLAGS="$F" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1 || { echo "[$F] build failed"; continue; }
  for cfg in "${CF[@]}"; do for rep in $(seq $REPS); do
    python bench.py --no-cpu-baseline --check 0 --steps 8 $cfg 2>/dev/null | line "staged $F" "$cfg"
  done; done
done
