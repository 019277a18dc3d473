#!/bin/bash
# same box, several builds: sweep one -D knob.  usage: gpu_solo_sweep.sh MACRO v1 v2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
M=$1; shift
for S in "$@"; do
  WGA_EXTRA_FLAGS="-D${M}=${S}u" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
  for cfg in "" "--records 10000 --mean-ops 50000" "--records 1000000 --mean-ops 500"; do
    for rep in 1 2; do
      python bench.py --no-cpu-baseline --check 0 --steps 8 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$M=$S [$cfg] K2 %.3f ms frac %.3f' % (d['kernel_ms']['k_paf2maf_expand'], d['roofline']['frac']))"
    done
  done
done
