"""wgatools_amd — MI355X (gfx950) engine for wgatools' CIGAR-driven hot path.

csrc/      hand-written HIP kernels + the C-ABI (include/wga_hip.h) -> libwgahip.so
_lib.py    ctypes binding of the C-ABI (no fallback: raises without the library / a GPU)
engine.py  device arrays + one method per entry point
pipeline.py batch drivers (paf2maf+stat) over HBM-resident data
synth.py   synthetic workloads of BASELINE.json's shapes
"""
__version__ = "0.1.0"
