#!/bin/bash
# round 6: K5 at configs[3]'s stated size — window size / waves per block / pieces ahead, A/B builds (one process each)
mkdir -p gpurun_out/r06u
export K5_MODE=fused K5_REPS=3
for v in product k5d1 k5win14 k5win14d1; do
  echo "== $v"
  if [ $v = product ]; then unset WGA_LIB; else export WGA_LIB=build_variants/libwgahip_$v.so; fi
  timeout 400 python scripts/gpu_k5_scaling.py 10 2>&1 | grep -E "accumulate_final|fused ==" | tee -a gpurun_out/r06u/$v.txt
done
