#!/bin/bash
# round 4: the MAF walks (K3 / K4) with sixteen columns per bit mask — parity, then rates beside round 3's
set -u
mkdir -p gpurun_out/r04f
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -x -q -m gpu -k "maf" > gpurun_out/r04f/pytest.txt 2>&1
tail -4 gpurun_out/r04f/pytest.txt
timeout 600 python scripts/gpu_maf_kernels.py 2000000 1500 > gpurun_out/r04f/maf_kernels.txt 2>&1
cat gpurun_out/r04f/maf_kernels.txt | grep -v amdgpu
timeout 600 python scripts/gpu_maf_kernels.py 100000 15000 2>&1 | grep -v amdgpu | tee gpurun_out/r04f/maf_kernels_15k.txt
