"""Where does the run-to-run spread of k_paf2maf_expand come from?  One process: the same batch, the output buffer
re-allocated several times (and shifted by a few KB), K2 timed per step with the library's HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wgatools_amd import engine, synth, pipeline

dev = torch.device("cuda", 0)
eng = engine.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
EARLY = os.environ.get("EARLY_OUT", "0") != "0"
early = torch.empty(17 * 10 ** 9, dtype=torch.uint8, device=dev) if EARLY else None   # before anything else touches HBM
tb = synth.make_paf_batch_torch(0x5747415F + 2, 100_000, 5000, int(sys.argv[1]) * 1_000_000 if len(sys.argv) > 1 else 50_000_000, dev)
eng.set_param("expand_timing", 1)

def raw_rates(buf):
    """plain device rates on the same bytes: fill (write only) and copy (read + write), GB/s"""
    n = buf.numel() // 2 // 4096 * 4096
    a, b = buf[:n], buf[n:2 * n]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    a.fill_(1); b.copy_(a); torch.cuda.synchronize()
    e[0].record()
    for _ in range(3): buf.fill_(7)
    e[1].record()
    for _ in range(3): b.copy_(a)
    e[2].record(); torch.cuda.synchronize()
    return 3 * buf.numel() / e[0].elapsed_time(e[1]) / 1e6, 3 * 2 * n / e[1].elapsed_time(e[2]) / 1e6

def run(job, steps):
    out = []
    for _ in range(steps):
        eng.expand_timing()
        job.stat(); job.layout(); job.expand()
        torch.cuda.synchronize()
        ms, n = eng.expand_timing()
        out.append(ms / max(n, 1))
    return out

hold = []
if EARLY:
    job = pipeline.Paf2MafStatJob(eng, tb, out=early)
    job.bind_stream()
    t = run(job, 12)
    print("early buffer: out @ 0x%x  K2 ms: %s  | mean of last 8 = %.3f" % (job.out.data_ptr(), " ".join("%.2f" % x for x in t), sum(t[4:]) / 8))
    for off in (4096, 65536, 1 << 20, (1 << 21) + 4096, 1 << 26, (1 << 30) + 12345 * 64):
        job = pipeline.Paf2MafStatJob(eng, tb, out=early[off:])
        job.bind_stream()
        t = run(job, 8)
        print("early buffer + %d: K2 ms mean of last 4 = %.3f" % (off, sum(t[4:]) / 4))
    # the same bytes of HBM, but the batch's other buffers (counts, offsets, tile workspace) allocated anew
    sys.exit(0)
for trial in range(6):
    job = pipeline.Paf2MafStatJob(eng, tb)
    job.bind_stream()
    t = run(job, 12)
    fr, cr = raw_rates(job.out)
    print("alloc %d: out @ 0x%x  K2 ms mean of last 8 = %.3f | torch fill %.0f GB/s, copy %.0f GB/s on the same buffer" % (trial, job.out.data_ptr(), sum(t[4:]) / 8, fr, cr))
    if trial % 2 == 0:
        hold.append(job.out)            # keep this allocation alive: the next one lands elsewhere
    del job
    torch.cuda.empty_cache()
job = pipeline.Paf2MafStatJob(eng, tb)
job.bind_stream()
t = run(job, 60)
print("60 steps on one allocation: first 10 %s ... last 10 %s" % (" ".join("%.2f" % x for x in t[:10]), " ".join("%.2f" % x for x in t[-10:])))
