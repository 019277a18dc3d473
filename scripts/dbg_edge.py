import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo")
import numpy as np
import torch  # noqa
from wgatools_amd import build as b, engine, _lib, synth
import parity_cases as pc
eng = engine.Engine(0, _lib.load(b.HIP_LIB))
bt = pc.edge_case_batch(eng)
n = len(bt["strand_neg"])
rng = np.random.default_rng(5)
which = sys.argv[1]
if which == "a":
    pc.check_paf2maf(eng, bt); print("a ok", flush=True)
elif which == "b":
    pc.check_paf2maf(eng, bt, pre=(rng.integers(0, 33, n), rng.integers(0, 33, n), rng.integers(0, 3, n))); print("b ok", flush=True)
elif which == "c":
    pc.check_paf2maf(eng, bt, force_slow=1); print("c ok", flush=True)
eng.close()
