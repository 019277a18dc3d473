#!/bin/bash
# rocprofv3 kernel stats + PMC passes of the bench command (no tests): usage gpu_prof_only.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --no-extras --check 0 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-extras --no-placement-probe --check 0 > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $R; find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
bash scripts/gpu_pmc.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1; tail -2 gpurun_out/${TAG}_pmc.log | cut -c1-300
python -c "import json; d=json.load(open('$OUT/bench_plain.json')); print('plain bench K2 ms', d['kernel_ms']['k_paf2maf_expand'], 'frac', d['roofline']['frac'])"
python -c "import json; d=json.load(open('$OUT/prof_bench.json')); print('under rocprof K2 ms', d['kernel_ms']['k_paf2maf_expand'])"
