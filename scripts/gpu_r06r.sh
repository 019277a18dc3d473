#!/bin/bash
# round 6: K9 (BED text) staged through LDS — its test and its time
mkdir -p gpurun_out/r06r
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pafcov_format or pafcov_cli or bed" > gpurun_out/r06r/tests.txt 2>&1
tail -3 gpurun_out/r06r/tests.txt
python scripts/gpu_other_kernels.py > gpurun_out/r06r/other.txt 2>&1
grep "K9\|K10\|K11" gpurun_out/r06r/other.txt
