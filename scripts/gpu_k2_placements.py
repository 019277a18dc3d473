"""Row kernels over several placements of the output buffer in one process: does the kernel's time follow the HBM region the
15 GB arena lies in (profiles/r02_k2_experiments.md sections 7-10) — for v1 (expand_variant 0) it does by +-8 %.
  python scripts/gpu_k2_placements.py [pool_mb] [lib:params ...]     (lib = tree | build_variants name)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wgatools_amd import build, engine, synth, pipeline, _lib

dev = torch.device("cuda", 0)
pool_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cands = sys.argv[2:] or ["tree:expand_variant=0", "tree:expand_variant=3"]
tb = synth.make_paf_batch_torch(0x5747415F + 2, 100_000, 5000, pool_mb * 1_000_000, dev)
engs = []
for c in cands:
    name, _, ps = c.partition(":")
    path = build.HIP_LIB if name == "tree" else os.path.join(build.ROOT, "build_variants", "libwgahip_%s.so" % name)
    eng = engine.Engine(0, _lib.load(path, require_all=(name == "tree")))
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for p in ps.split(","):
        if p:
            eng.set_param(p.split("=")[0], int(p.split("=")[1]))
    eng.set_param("expand_timing", 1)
    eng.set_param("expand_drain_min", 64)
    engs.append((c, eng))
hold = []
rows = []
for trial in range(8):
    out = None
    line = []
    for c, eng in engs:
        job = pipeline.Paf2MafStatJob(eng, tb, out=out)
        out = job.out
        job.bind_stream()
        job.stat(); job.layout(); job.expand(); torch.cuda.synchronize()
        eng.expand_timing()
        for _ in range(6):
            job.expand()
        torch.cuda.synchronize()
        ms, n = eng.expand_timing()
        line.append(ms / n)
        del job
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    out.fill_(7); torch.cuda.synchronize()
    e[0].record()
    for _ in range(3): out.fill_(7)
    e[1].record(); torch.cuda.synchronize()
    fill = 3 * out.numel() / e[0].elapsed_time(e[1]) / 1e6
    print("alloc %d @ 0x%x: " % (trial, out.data_ptr()) + "  ".join("%.3f" % x for x in line) + "  ms | torch fill %.0f GB/s" % fill, flush=True)
    rows.append(line)
    hold.append(out)     # keep every allocation: the next one lands elsewhere (8 x 15 GB)
print("candidates:", cands)
for k, (c, _e) in enumerate(engs):
    v = sorted(r[k] for r in rows)
    print("  %-50s min %.3f  median %.3f  max %.3f" % (c, v[0], v[len(v) // 2], v[-1]))
