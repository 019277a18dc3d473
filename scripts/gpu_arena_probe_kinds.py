"""which plain write pattern ranks output buffers the way the row kernel (K2) does?  One process, NC candidate buffers of
the configs[1] output size; on every candidate: K2's time on the real batch, then the rate of every probe kind
(wga_arena_probe).  Prints the table and the rank correlation of every kind with K2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wgatools_amd import engine, synth, pipeline

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 12
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
tb = synth.make_paf_batch_torch(0x5747415F + 2, 100_000, 5000, 50_000_000, dev)
eng = engine.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_param("expand_timing", 1)
eng.set_param("expand_variant", variant)
eng.set_param("expand_drain_min", 64)
job0 = pipeline.Paf2MafStatJob(eng, tb)
nbytes = int(job0.out.numel())
del job0
torch.cuda.empty_cache()
cands = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(nc)]
rows = []
for k, buf in enumerate(cands):
    job = pipeline.Paf2MafStatJob(eng, tb, out=buf)
    job.bind_stream()
    job.stat(); job.layout()
    for _ in range(2): job.expand()
    torch.cuda.synchronize(); eng.expand_timing()
    for _ in range(4): job.expand()
    torch.cuda.synchronize()
    ms, n = eng.expand_timing()
    k2 = ms / n
    del job
    kinds = [0] + list(range(6, 14))
    rates = [eng.arena_probe(buf, nbytes, kind) for kind in kinds]
    rates2 = [eng.arena_probe(buf, nbytes, kind) for kind in kinds]
    rows.append([k2] + rates + rates2)
    print("cand %2d @0x%x  K2 %.3f ms | copy %.0f xcd-local fill, rotation 0..7: %s | again %s" % (
        k, buf.data_ptr(), k2, rates[0], " ".join("%.0f" % x for x in rates[1:]), " ".join("%.0f" % x for x in rates2[1:])), flush=True)
a = np.array(rows)
def ranks(x): return np.argsort(np.argsort(x)).astype(float)
for j, name in enumerate(["copy"] + ["xcd rot %d" % r for r in range(8)]):
    r = np.corrcoef(ranks(a[:, 0]), ranks(-a[:, 1 + j]))[0, 1]
    pick = int(np.argmax(a[:, 1 + j]))
    print("%-9s rank correlation with K2 %+.2f; its pick gives K2 %.3f ms (best %.3f, median %.3f, worst %.3f); repeatability %.3f" % (
        name, r, a[pick, 0], a[:, 0].min(), np.median(a[:, 0]), a[:, 0].max(), np.corrcoef(a[:, 1 + j], a[:, 10 + j])[0, 1]))
