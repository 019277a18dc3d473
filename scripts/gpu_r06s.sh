#!/bin/bash
# round 6: K11's fill with the per-block table (k_elem_blocks) — tests and times
mkdir -p gpurun_out/r06s
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or maf2paf or runs or bridge or roundtrip" > gpurun_out/r06s/tests.txt 2>&1
tail -3 gpurun_out/r06s/tests.txt
python scripts/gpu_other_kernels.py 2>&1 | grep "K11" | tee gpurun_out/r06s/other.txt
python scripts/gpu_maf_kernels.py 2>&1 | grep "K11\|blocks" | tee gpurun_out/r06s/maf.txt
python scripts/gpu_maf_kernels.py 2000000 1500 2>&1 | grep "K11\|blocks" | tee -a gpurun_out/r06s/maf.txt
