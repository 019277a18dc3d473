"""expand_drain_min against the size of the sequence pools: one process, one output buffer per shape"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wgatools_amd import engine, synth, pipeline
dev = torch.device("cuda", 0)
eng = engine.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_param("expand_timing", 1)
for pool in [int(x) for x in (sys.argv[1:] or "50 100 150 200 300 500 1000".split())]:
    tb = synth.make_paf_batch_torch(0x5747415F + 2, 100_000, 5000, pool * 1_000_000, dev)
    job = pipeline.Paf2MafStatJob(eng, tb)
    job.bind_stream(); job.stat(); job.layout(); job.expand(); torch.cuda.synchronize()
    res = []
    eng.set_param("expand_drain_min", 0)
    eng.set_param("expand_autotune", 1)       # forget what the previous shape settled on
    for _ in range(6): job.expand()
    torch.cuda.synchronize()
    chosen = eng.get_param("expand_drain_min"), eng.get_param("expand_autotune_settled")
    for dm in (16, 24, 32, 48, 64, 0):
        eng.set_param("expand_drain_min", dm)
        eng.expand_timing()
        for _ in range(6): job.expand()
        torch.cuda.synchronize()
        ms, n = eng.expand_timing()
        res.append("%s %.3f" % (dm if dm else "auto", ms / n))
    print("2 x %4d MB pools: K2 ms by drain_min: %s   (autotune chose %d, settled %d)" % (pool, " | ".join(res), chosen[0], chosen[1]))
    del job, tb
    torch.cuda.empty_cache()
