"""profiling helper: per-tile s_memtime stamps of k_paf2maf_expand (not part of the product)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from wgatools_amd import engine, pipeline, synth
dev = torch.device("cuda", 0)
eng = engine.Engine(0)
tb = synth.make_paf_batch_torch(0x5747415F + 2, 100_000, 5000, 50_000_000, dev)
job = pipeline.Paf2MafStatJob(eng, tb)
job.bind_stream()
nt = (tb["n_ops"] + 1023) // 1024
dbg = torch.zeros(nt * 8, dtype=torch.int64, device=dev)
job.step(); torch.cuda.synchronize()
eng.set_param("expand_dbg_ptr", dbg.data_ptr())
job.expand(); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 8)
t0 = d[:, 0]
names = ["phaseA", "setup", "row_t", "row_q", "rest"]
segs = [d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 4] - d[:, 3], d[:, 5] - d[:, 4]]
ok = (d[:, 6] == 1) & (d[:, 4] > 0)
print("tiles", len(d), "single-segment tiles", ok.sum(), "s_memtime ticks (100 MHz => 10 ns)")
for n, x in zip(names, segs):
    x = x[ok]
    print("%-8s mean %8.1f  p50 %8.1f  p90 %8.1f" % (n, x.mean(), np.median(x), np.percentile(x, 90)))
tot = (d[:, 5] - d[:, 0])[ok]
print("total    mean %8.1f  p50 %8.1f  p90 %8.1f" % (tot.mean(), np.median(tot), np.percentile(tot, 90)))
print("kernel span ticks", d[:, 5].max() - d[:, 0].min())
