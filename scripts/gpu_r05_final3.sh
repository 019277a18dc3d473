#!/bin/bash
# round 5, last call: the GPU suite and smoke on the final tree, the bench line (plain, and with the process group forced:
# the RCCL branches of the N > 1 path at world size 1), the MAF kernels
TAG=${1:-r05final3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "config 4 at size|passed|failed" $OUT/pytest_gpu.log | tail -5
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_n1.json
timeout 600 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err; echo "force-dist rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_force_dist.json').readline()); print('ranks_seen_by_rccl', d['config'].get('ranks_seen_by_rccl'), 'value %.3e' % d['value'], 'scaling', d['scaling'])"
timeout 300 python scripts/gpu_maf_kernels.py 2>&1 | grep -E "^K[34]|blocks" | tee $OUT/maf_kernels.log
