#!/bin/bash
# end to end at configs[1] size (100 000 records, 1.2 GB of PAF text -> 15 GB of MAF), then the bench line with its e2e leg
TAG=${1:-r03j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
df -h /tmp | tail -1
python -c "from wgatools_amd import build; print(build.build_cli())"
timeout 1500 python scripts/gpu_cli_e2e.py ${2:-100000} /tmp/wga_e2e 100000 > $OUT/cli_e2e.txt 2>&1; echo "e2e rc=$?"; cat $OUT/cli_e2e.txt | tail -60
rm -rf /tmp/wga_e2e
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
r = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "roofline", "e2e") if k in r}, indent=1)[:3500])
PY
tail -5 $OUT/bench.err
