#!/bin/bash
# round 4: pafpseudo's rows (both modes) through the streaming row kernel — parity, then the call's time beside the block kernel's
set -u
mkdir -p gpurun_out/r04d
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pafpseudo or stream_kernel" > gpurun_out/r04d/pytest.txt 2>&1
tail -5 gpurun_out/r04d/pytest.txt
timeout 600 python scripts/gpu_other_kernels.py 100000 5000 > gpurun_out/r04d/other_kernels.txt 2>&1
grep -E "K6" gpurun_out/r04d/other_kernels.txt
timeout 300 python scripts/gpu_other_kernels.py 1000000 500 > gpurun_out/r04d/other_kernels_500.txt 2>&1
grep -E "K6" gpurun_out/r04d/other_kernels_500.txt
for v in sym5 sym7 sym8; do
  if [ -f build_variants/libwgahip_$v.so ]; then
    echo "== $v"
    WGA_LIB_VARIANT=$v timeout 300 python scripts/gpu_other_kernels.py 100000 5000 2>&1 | grep -E "K6 pafpseudo symbol"
  fi
done
