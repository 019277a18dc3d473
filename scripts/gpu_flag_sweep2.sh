#!/bin/bash
# same box, several builds; bench configs given in CFGS (newline separated)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for F in "$@"; do
  WGA_EXTRA_FLAGS="$F" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
  while IFS= read -r cfg; do
    for rep in 1 2; do
      python bench.py --no-cpu-baseline --no-extras --check 0 --steps 6 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$F] [$cfg] K2 %.3f ms frac %.3f' % (d['kernel_ms']['k_paf2maf_expand'], d['roofline']['frac']))"
    done
  done <<< "$CFGS"
done
