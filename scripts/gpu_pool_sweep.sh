#!/bin/bash
# K2 against the size of the sequence pools and of the resident batch (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for cfg in "--pool-mb 50" "--pool-mb 200" "--pool-mb 1000" "--pool-mb 3000" "--records 10000 --mean-ops 50000 --pool-mb 50" "--records 10000 --mean-ops 50000 --pool-mb 1000" "--records 40000 --mean-ops 50000 --pool-mb 50" "--records 40000 --mean-ops 50000 --pool-mb 1000"; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --check 0 --steps 6 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-52s %.3e ops/s  step %.2f ms  K2 %.2f ms  frac %.3f  K1 %.2f ms  (%.2e ops, %.1f GB out)' % ('$cfg', d['value'], d['ms_per_step'], d['kernel_ms']['k_paf2maf_expand'], d['roofline']['frac'], d['kernel_ms']['k_cigar_stat'], d['config']['ops_per_gpu'], d['config']['output_bytes_per_gpu']/1e9))"
done
