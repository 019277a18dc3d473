#!/bin/bash
# round 4, call 1: the LDS-DMA streaming micro-benchmark (is a ring-staged row at the rate of a plain copy?) and the counter
# pass on K5's list / replay kernels that round 3 left open.
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== micro: stream_dma_copy"
( cd scripts/micro && { [ -x stream_dma_copy ] || hipcc --offload-arch=gfx950 -O3 -o stream_dma_copy stream_dma_copy.hip 2>/dev/null; } && timeout 300 ./stream_dma_copy ) > $OUT/stream_dma_copy.txt 2>&1; echo "micro rc=$?"; cat $OUT/stream_dma_copy.txt
echo "== K5 list pass: counters (SQ mix, waits, TCC requests) over one 10 GB call"
WGA_PMC_CMD="python $R/scripts/gpu_k5_scaling.py 1" timeout 500 bash scripts/gpu_pmc.sh ${TAG}_k5pmc "sq1 sq2 sq3 tcc fetch write" 2>&1 | grep -E "k_cov|rc=" | cut -c1-400 | tee $OUT/k5_pmc.log
