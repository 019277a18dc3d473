"""K10 paf2chain (k_cigar_chain): count and fill pass on BASELINE configs[1]'s batch and on two other record lengths.
usage: python scripts/gpu_k10.py [records mean_ops]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wgatools_amd import engine, synth

dev = torch.device("cuda", 0)
eng = engine.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)

def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

shapes = [(100_000, 5000), (10_000, 50_000), (1_000_000, 500)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[k]), int(sys.argv[k + 1])) for k in range(1, len(sys.argv) - 1, 2)]
for nrec, mean in shapes:
    tb = synth.make_paf_batch_torch(0x5747415F + 2, nrec, mean, 50_000_000, dev)
    batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], tb["n"], tb["n_ops"])
    n, n_ops = tb["n"], tb["n_ops"]
    trim = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    nb = torch.zeros(n, dtype=torch.int64, device=dev)
    dg = torch.zeros((n, 3), dtype=torch.int64, device=dev)
    ms_c = timed(lambda: eng.cigar_chain(batch, trim=trim, nbytes=nb, diag=dg))
    if os.environ.get("K10_DBG"):
        ser = dg[:, 1] != -1
        print("   serial path: %d of %d records; first steps %s" % (int(ser.sum()), n, dg[ser, 1][:8].tolist()))
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    eng.exclusive_scan_u64(n, nb, off)
    tot = int(off[-1].item())
    txt = torch.zeros(tot + 8, dtype=torch.uint8, device=dev)
    ms_f = timed(lambda: eng.cigar_chain(batch, out=txt, out_off=off))
    print("K10 paf2chain %d x %d op: count pass %.3f ms %.0f GB/s (4 B/op); fill pass %.3f ms %.0f GB/s (4 B/op + %.2f GB of text); checksum %d" % (
        n, mean, ms_c, 4 * n_ops / ms_c / 1e6, ms_f, (4 * n_ops + tot) / ms_f / 1e6, tot / 1e9,
        int(txt.to(torch.int64).sum().item())))
    del tb, batch, txt
