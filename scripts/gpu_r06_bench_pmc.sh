#!/bin/bash
# the traffic passes of the row kernel (FETCH / WRITE) on the current source, then the driver's bench command plain and under rocprofv3
TAG=${1:-r06bp}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/gpu_pmc.sh ${TAG}_pmc "sq1 sq2 fetch write" 2>&1 | grep -E "rc=" 
python - <<PY
import csv, glob
n = 0
for f in glob.glob("$R/gpurun_out/${TAG}_pmc/fetch/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("k_paf2maf_expand_s") and row["Counter_Name"] == "FETCH_SIZE": n += 1
print("launches", n)
import subprocess, sys
subprocess.check_call([sys.executable, "scripts/pmc_summarise.py", "gpurun_out/${TAG}_pmc", "gpurun_out/$TAG/r06", str(n), "100000", "5000", "499238357"])
PY
cp $OUT/r06_pmc_traffic.json profiles/r06_pmc_traffic.json   # on the box: the bench below reads it (same source hash)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
cut -c1-300 $OUT/bench_n1.json
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extras > $OUT/bench_n1_under_rocprof.json 2> $OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c 'head -4 {}' | cut -c1-160
find $OUT/prof -name '*kernel_trace.csv' -delete
