#!/bin/bash
# round 6: K5 with 16-byte descriptors made by the placement, marks through per-lane dummy words, 16 K windows, one piece ahead
mkdir -p gpurun_out/r06v
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pafcov and not stated" > gpurun_out/r06v/tests.txt 2>&1
tail -3 gpurun_out/r06v/tests.txt
export K5_MODE=fused K5_REPS=3
for v in product k5d2 k5d0; do
  echo "== $v"
  if [ $v = product ]; then unset WGA_LIB; else export WGA_LIB=build_variants/libwgahip_$v.so; fi
  timeout 400 python scripts/gpu_k5_scaling.py 10 2>&1 | grep -E "accumulate_final|fused ==" | tee -a gpurun_out/r06v/$v.txt
done
unset WGA_LIB
K5_MODE=both K5_REPS=2 timeout 400 python scripts/gpu_k5_scaling.py 10 2>&1 | grep -E "accumulate|fused ==" | tee gpurun_out/r06v/both.txt
