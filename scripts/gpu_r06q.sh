#!/bin/bash
# round 6: counter passes of the MAF stream kernels (200 000 x 1 500 columns): instruction mix, waits, FETCH / WRITE
TAG=${1:-r06q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
WGA_PMC_CMD="python $R/scripts/gpu_maf_kernels.py" timeout 1500 bash scripts/gpu_pmc.sh ${TAG}_mafpmc "sq1 sq2 fetch write" 2>&1 | grep -E "k_maf_stream|rc=" | cut -c1-420
