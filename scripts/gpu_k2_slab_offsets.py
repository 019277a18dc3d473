"""K2 against the position of its output buffer inside ONE 208 GB allocation (profiles/r02_k2_experiments.md, sections 7 and 9):
is the level a function of the physical position?  drain_min fixed at 64, same batch, 13 positions, twice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wgatools_amd import engine, synth, pipeline
dev = torch.device("cuda", 0)
G = 1024 ** 3
slab = torch.empty(208 * G, dtype=torch.uint8, device=dev)        # first: the rest of the process allocates around it
eng = engine.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
tb = synth.make_paf_batch_torch(0x5747415F + 2, 100_000, 5000, 50_000_000, dev)
eng.set_param("expand_timing", 1)
eng.set_param("expand_drain_min", 64)
for rep in range(2):
    row = []
    for off in range(0, 193, 16):
        job = pipeline.Paf2MafStatJob(eng, tb, out=slab[off * G:])
        job.bind_stream(); job.stat(); job.layout()
        for _ in range(2): job.expand()
        torch.cuda.synchronize(); eng.expand_timing()
        for _ in range(4): job.expand()
        torch.cuda.synchronize()
        ms, n = eng.expand_timing()
        row.append("%d:%.2f" % (off, ms / n))
        del job
    print("K2 ms by offset of the output buffer in the slab (GB:ms)  " + "  ".join(row))
