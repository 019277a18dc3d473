#!/bin/bash
# round 5: K18 (BGZF deflate on the device) — parity on the GPU, at size, through the command line; the bench line with the
# `.gz` leg and the counter passes' traffic; the commands at size after they leave without tearing down
TAG=${1:-r05k18}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "bgzf" > $OUT/pytest_bgzf.log 2>&1; echo "pytest bgzf rc=$?"; grep -E "K18 at size|passed|failed|Error" $OUT/pytest_bgzf.log | tail -6
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -k "paf2maf_end_to_end or gz_outputs or call_readme or stat_paf_fixture or pafcov_fixture" > $OUT/pytest_cli.log 2>&1; echo "pytest cli rc=$?"; tail -3 $OUT/pytest_cli.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('$OUT/bench_n1.json').readline())
print('value %.3e frac %.3f traffic %s' % (d['value'], d['roofline']['frac'], d['roofline']['traffic']))
for k in ('stat','paf2maf','paf2maf_gz'):
    e=d['e2e'].get(k,{}); print(k, {x:e.get(x) for x in ('wall_s','output_bytes','ratio','members','check','error')}); print('   ', e.get('phases'))
PY
timeout 900 python scripts/gpu_e2e_at_size.py > $OUT/e2e_at_size.log 2>&1; echo "e2e rc=$?"; grep -A1 -E "^call_maf|^stat_maf|^maf2paf|^pafcov|^stat_paf" $OUT/e2e_at_size.log | cut -c1-420
cp gpurun_out/e2e_at_size.json $OUT/ 2>/dev/null
