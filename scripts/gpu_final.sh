#!/bin/bash
# the round's evidence in one call: GPU tests + smoke + bench + rocprof stats (gpu_round.sh), PMC passes, command-line
# timings, secondary kernels.  usage: gpurun -- 'bash scripts/gpu_final.sh r02'
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/gpu_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1; grep -E "passed|failed|rc=" gpurun_out/${TAG}_round.log | head -8
bash scripts/gpu_pmc.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/${TAG}_pmc.log
bash scripts/gpu_e2e_timing.sh > gpurun_out/${TAG}_e2e.txt 2>&1; tail -12 gpurun_out/${TAG}_e2e.txt
python scripts/gpu_maf_kernels.py > gpurun_out/${TAG}_maf_kernels.txt 2>&1; tail -5 gpurun_out/${TAG}_maf_kernels.txt
python scripts/gpu_other_kernels.py > gpurun_out/${TAG}_other_kernels.txt 2>&1; tail -12 gpurun_out/${TAG}_other_kernels.txt
