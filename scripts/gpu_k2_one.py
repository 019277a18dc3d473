"""K2 alone on BASELINE configs[1]'s batch with a given library build: python scripts/gpu_k2_one.py LIB [records mean_ops pool_mb]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wgatools_amd import engine, synth, pipeline, _lib
lib = _lib.load(sys.argv[1])
rec, mean, pool = (int(x) for x in (sys.argv[2:5] if len(sys.argv) >= 5 else (100000, 5000, 50)))
dev = torch.device("cuda", 0)
tb = synth.make_paf_batch_torch(0x5747415F + 2, rec, mean, pool * 1_000_000, dev)
eng = engine.Engine(0, lib)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_param("expand_timing", 1)
job = pipeline.Paf2MafStatJob(eng, tb)
job.bind_stream()
job.step(); torch.cuda.synchronize(); eng.expand_timing()
for _ in range(4):
    job.expand()
torch.cuda.synchronize()
ms, n = eng.expand_timing()
print("K2 %.3f ms (%d launches) variant %d" % (ms / n, n, eng.get_param("expand_variant")))
eng.close()
