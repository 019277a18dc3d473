#!/bin/bash
# round 4: rocprofv3 --kernel-trace --stats over the secondary kernels' scripts (K5-K12 on configs[1]'s batch, K3 / K4 on configs[2]'s shape)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04m
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/other -o p -- python $R/scripts/gpu_other_kernels.py 100000 5000 > $OUT/other.log 2>&1; echo "other rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/maf -o p -- python $R/scripts/gpu_maf_kernels.py 2000000 1500 > $OUT/maf.log 2>&1; echo "maf rc=$?"
find $OUT -name '*kernel_trace.csv' -delete
for d in other maf; do f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1); echo "== $d"; grep -E '^"k_' $f | cut -d, -f1-4 | cut -c1-150 | head -30; done
