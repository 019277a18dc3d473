"""profiling helper: per-wave s_memtime sums of the window kernel's phases (a -DWGA_PROFILE build; not part of the product)
usage: python scripts/gpu_k2w_stamps.py LIB [records mean_ops pool_mb]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from wgatools_amd import engine, pipeline, synth, _lib
lib = _lib.load(sys.argv[1])
rec, mean, pool = (int(x) for x in (sys.argv[2:5] if len(sys.argv) >= 5 else (100000, 5000, 50)))
dev = torch.device("cuda", 0)
eng = engine.Engine(0, lib)
eng.set_param("expand_variant", 2)
tb = synth.make_paf_batch_torch(0x5747415F + 2, rec, mean, pool * 1_000_000, dev)
job = pipeline.Paf2MafStatJob(eng, tb)
job.bind_stream()
nt = (tb["n_ops"] + 1023) // 1024
dbg = torch.zeros(nt * 4 * 8, dtype=torch.int64, device=dev)
job.step(); torch.cuda.synchronize()
eng.set_param("expand_dbg_ptr", dbg.data_ptr())
job.expand(); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
names = ["phase A (to barrier)", "whole tile", "w: record+search+views", "w: table", "w: offsets+issue", "w: wait+stage", "w: fix-ups", "between windows / plan"]
print("waves", len(d), "ticks of s_memtime (100 MHz: 10 ns)")
for k, n in enumerate(names):
    x = d[:, k]
    print("%-26s mean %8.1f  p50 %8.1f  p90 %8.1f  sum/waves %8.1f" % (n, x.mean(), np.median(x), np.percentile(x, 90), x.sum() / len(d)))
rest = d[:, 1] - d[:, 0] - d[:, 2:].sum(1)
print("%-26s mean %8.1f  p50 %8.1f  p90 %8.1f" % ("w: copy-out + wave end", rest.mean(), np.median(rest), np.percentile(rest, 90)))
