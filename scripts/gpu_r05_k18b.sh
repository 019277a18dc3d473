#!/bin/bash
# K18 after the rank-by-compaction / slicing-by-4 / word-wise packing changes: parity on the GPU again, at size, and through the command line
TAG=${1:-r05k18b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -s -k "bgzf" > $OUT/pytest_bgzf.log 2>&1; echo "pytest bgzf rc=$?"; grep -E "K18 at size|passed|failed|Error" $OUT/pytest_bgzf.log | tail -6
timeout 60 python -m pytest tests/test_gpu_cli.py -x -q -k "paf2maf_end_to_end or gz_outputs" > $OUT/pytest_cli.log 2>&1; echo "pytest cli rc=$?"; tail -2 $OUT/pytest_cli.log
exit 0
