"""K1 (wga_cigar_stat) alone on BASELINE configs[1]'s batch (100 000 records x mean 5 kop): ms per launch and TB/s of op stream,
for the product build and the A/B builds named on the command line (build_variants/libwgahip_<name>.so)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wgatools_amd import engine, synth, _lib, build

dev = torch.device("cuda", 0)
tb = synth.make_paf_batch_torch(0x5747415F + 2, 100_000, 5000, 50_000_000, dev)
n, n_ops = tb["n"], tb["n_ops"]
for name in [None] + sys.argv[1:]:
    lib = _lib.load(os.path.join(build.ROOT, "build_variants", "libwgahip_%s.so" % name)) if name else None
    eng = engine.Engine(0, lib)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], n, n_ops)
    counts = torch.zeros((n, 11), dtype=torch.int64, device=dev)
    diag = torch.zeros((n, 3), dtype=torch.int64, device=dev)
    tiles = torch.zeros((n_ops // 1024 + 2) * 88 + 16, dtype=torch.uint8, device=dev)
    f = lambda: eng.cigar_stat(batch, counts, diag, tiles)
    f(); torch.cuda.synchronize()
    best = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / 10)
    ms = sorted(best)[len(best) // 2]
    print("K1 %-14s %.3f ms per call (memsets + tile records + walk) = %.2f TB/s of op stream; checksum %d" % (
        name or "product", ms, 4 * n_ops / ms / 1e9, int(counts.sum().item())), flush=True)
    eng.close()
