"""one two-row block of 10^8 columns through the MAF walks: the count calls, and the fill calls behind
them (the two-call protocol); 1 % SNP, 0.15 % indel-open as scripts/gpu_maf_kernels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wgatools_amd import engine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
t = alpha[torch.randint(0, 4, (L,), device=dev, generator=g)]
q = t.clone()
snp = torch.rand(L, device=dev, generator=g) < 0.012
q[snp] = alpha[torch.randint(0, 4, (int(snp.sum()),), device=dev, generator=g)]
opn = torch.rand(L, device=dev, generator=g) < 0.0015
idx = torch.arange(L, device=dev)
t[opn & (idx % 2 == 0)] = 45
q[opn & (idx % 2 == 1)] = 45
del snp, opn, idx
rows = torch.cat([t, q]).contiguous()
one = lambda v, dt=torch.int64: torch.tensor([v], dtype=dt, device=dev)
t_off, q_off, cols, strand = one(0), one(L), one(L), one(0, torch.uint8)
eng = engine.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
counts = torch.zeros((1, 11), dtype=torch.int64, device=dev)
run_cnt, run_off = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(2, dtype=torch.int64, device=dev)

def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

k3c = lambda: eng.maf_pair_stat(1, rows, t_off, q_off, cols, strand, counts=counts, run_cnt=run_cnt)
ms = timed(k3c)
print("K3 count call          : %.3f ms  %.0f GB/s (2 B/column)" % (ms, 2 * L / ms / 1e6))
eng.exclusive_scan_u64(1, run_cnt, run_off)
nr = int(run_off[-1].item())
runs = torch.zeros(nr + 1, dtype=torch.int64, device=dev)
k3f = lambda: eng.maf_pair_stat(1, rows, t_off, q_off, cols, strand, counts=counts, run_cnt=run_cnt, runs=runs, run_off=run_off)
ms2 = timed(lambda: (k3c(), k3f())) - ms
print("K3 fill call behind it : %.3f ms  %.0f GB/s (2 B/column + 8 B/run, %d runs)" % (ms2, (2 * L + 8 * nr) / ms2 / 1e6, nr))
ms3 = timed(k3f)
print("K3 fill call on its own: %.3f ms  (lists and walks the pieces twice)" % ms3)
k4c = lambda: eng.maf_call_runs(1, rows, t_off, q_off, cols, run_cnt=run_cnt)
ms = timed(k4c)
print("K4 count call          : %.3f ms  %.0f GB/s" % (ms, 2 * L / ms / 1e6))
eng.exclusive_scan_u64(1, run_cnt, run_off)
nr = int(run_off[-1].item())
cruns = torch.zeros(3 * nr + 3, dtype=torch.int64, device=dev)
k4f = lambda: eng.maf_call_runs(1, rows, t_off, q_off, cols, run_cnt=run_cnt, runs=cruns, run_off=run_off)
ms2 = timed(lambda: (k4c(), k4f())) - ms
print("K4 fill call behind it : %.3f ms  %.0f GB/s (2 B/column + 24 B/run, %d runs)" % (ms2, (2 * L + 24 * nr) / ms2 / 1e6, nr))
print("one block of %.1e columns" % L)
