#!/bin/bash
# the round's last seconds of GPU: the bench line of the final tree (with the `.gz` leg), then the command-line suite for as long as it fits
TAG=${1:-r05last}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 75 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('$OUT/bench_n1.json').readline())
print('value %.3e frac %.3f traffic %s' % (d['value'], d['roofline']['frac'], d['roofline']['traffic']))
for k in ('paf2maf','paf2maf_gz'):
    e=d['e2e'].get(k,{}); print(k, {x:e.get(x) for x in ('wall_s','output_bytes','ratio','check','error')}); print('   ', e.get('phases'))
PY
timeout 80 python -m pytest tests/test_gpu_cli.py -x -q > $OUT/pytest_cli.log 2>&1; echo "pytest cli rc=$?"; tail -2 $OUT/pytest_cli.log
exit 0
