"""Does a plain copy run at the same rate everywhere in HBM?  Allocates N slabs of 2 GB (default 120 = 240 GB of the 288),
and times torch's fill_ and copy_ inside each slab (profiles/r02_k2_experiments.md, section 7: the row kernel's time depends
on where its output buffer lies)."""
import sys, torch
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
G = 2 * 1024 ** 3
slabs = [torch.empty(G, dtype=torch.uint8, device=dev) for _ in range(n)]
torch.cuda.synchronize()
def rate(f, nbytes, reps=4):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return reps * nbytes / a.elapsed_time(b) / 1e6
rows = []
for k, s in enumerate(slabs):
    h = G // 2
    fr = rate(lambda: s.fill_(3), G)
    cr = rate(lambda: s[h:].copy_(s[:h]), 2 * h)
    rows.append((k, s.data_ptr(), fr, cr))
for k, p, fr, cr in rows:
    print("slab %3d @ 0x%x  fill %5.0f GB/s  copy %5.0f GB/s" % (k, p, fr, cr))
fs = sorted(r[2] for r in rows); cs = sorted(r[3] for r in rows)
print("fill min / median / max: %.0f / %.0f / %.0f   copy: %.0f / %.0f / %.0f" % (fs[0], fs[len(fs) // 2], fs[-1], cs[0], cs[len(cs) // 2], cs[-1]))
