"""throughput of the MAF column-walk kernels (K3 stat / maf2paf runs, K4 call runs) on a config-3 shaped
synthetic input: n blocks x ~L columns, 1.2 % SNP, 0.15 % indel-open"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wgatools_amd import engine

n_all = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
n = min(n_all, 200_000)           # rows are generated for n blocks and repeated up to n_all
rep = max(1, n_all // n)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
cols = torch.full((n,), L, dtype=torch.int64, device=dev)
tot = n * L
alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
t = alpha[torch.randint(0, 4, (tot,), device=dev, generator=g)]
q = t.clone()
snp = torch.rand(tot, device=dev, generator=g) < 0.012
q[snp] = alpha[torch.randint(0, 4, (int(snp.sum()),), device=dev, generator=g)]
# indels: open with p = 0.0015, geometric length (mean 3): mark runs by a cumulative trick
opn = torch.rand(tot, device=dev, generator=g) < 0.0015
ln = torch.zeros(tot, dtype=torch.int32, device=dev)
ln[opn] = (torch.empty(int(opn.sum()), device=dev).geometric_(1 / 3.0, generator=g)).to(torch.int32)
idx = torch.arange(tot, device=dev)
start = torch.where(opn, idx, torch.zeros_like(idx))
last_start = torch.cummax(start, 0).values
last_len = ln[last_start]
in_gap = (idx - last_start < last_len) & (last_start > 0)
which = (last_start % 2 == 0)
t[in_gap & which] = 45
q[in_gap & ~which] = 45
rows = torch.cat([t.repeat(rep), q.repeat(rep)]).contiguous()
n, tot = n * rep, tot * rep
cols = torch.full((n,), L, dtype=torch.int64, device=dev)
t_off = (torch.arange(n, device=dev) * L).to(torch.int64)
q_off = t_off + tot
strand = (torch.rand(n, device=dev, generator=g) < 0.1).to(torch.uint8)
lib = None
if os.environ.get("WGA_LIB"):     # an A/B build (wgatools_amd.build.build_hip_variant)
    from wgatools_amd import _lib
    lib = _lib.load(os.environ["WGA_LIB"])
eng = engine.Engine(0, lib)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
if os.environ.get("WGA_MAF_GROUP"):
    eng.set_param("maf_group", int(os.environ["WGA_MAF_GROUP"]))     # blocks per wave of the stream kernels (0: by the batch)
counts = torch.zeros((n, 11), dtype=torch.int64, device=dev)
run_cnt = torch.zeros(n, dtype=torch.int64, device=dev)
run_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)

def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

def k3_count():
    eng.maf_pair_stat(n, rows, t_off, q_off, cols, strand, counts=counts, run_cnt=run_cnt)
def k4_count():
    eng.maf_call_runs(n, rows, t_off, q_off, cols, run_cnt=crun_cnt)
# the two-call protocol: a fill call is timed behind its count call (count + fill, minus the count call alone) — a fill call
# on arrays no count call has just seen walks the long blocks twice (piece totals, then the runs)
ms = timed(k3_count)
print("K3 counts only      : %.3f ms  %.0f GB/s (2 B/column)" % (ms, 2 * tot / ms / 1e6))
eng.exclusive_scan_u64(n, run_cnt, run_off)
nruns = int(run_off[-1].item())
runs = torch.zeros(nruns + 1, dtype=torch.int64, device=dev)
def k3_fill():
    eng.maf_pair_stat(n, rows, t_off, q_off, cols, strand, counts=counts, run_cnt=run_cnt, runs=runs, run_off=run_off)
ms2 = timed(lambda: (k3_count(), k3_fill())) - ms
print("K3 counts + run list: %.3f ms  %.0f GB/s (2 B/column + 8 B/run, %d runs)" % (ms2, (2 * tot + 8 * nruns) / ms2 / 1e6, nruns))
crun_cnt = torch.zeros(n, dtype=torch.int64, device=dev)
ms = timed(k4_count)
print("K4 count pass       : %.3f ms  %.0f GB/s" % (ms, 2 * tot / ms / 1e6))
crun_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
eng.exclusive_scan_u64(n, crun_cnt, crun_off)
ncr = int(crun_off[-1].item())
cruns = torch.zeros(3 * ncr + 3, dtype=torch.int64, device=dev)
def k4_fill():
    eng.maf_call_runs(n, rows, t_off, q_off, cols, run_cnt=crun_cnt, runs=cruns, run_off=crun_off)
ms2 = timed(lambda: (k4_count(), k4_fill())) - ms
print("K4 run list         : %.3f ms  %.0f GB/s (2 B/column + 24 B/run, %d runs)" % (ms2, (2 * tot + 24 * ncr) / ms2 / 1e6, ncr))
print("blocks %d x %d columns = %.2e columns; strand- %.1f %%" % (n, L, tot, 100 * float(strand.float().mean())))
if os.environ.get("WGA_MAF_K11", "1") != "0":      # maf2paf's cg:Z: text and maf2chain's ops from the K3 run list (K11)
    tcnt = torch.zeros(n, dtype=torch.int64, device=dev)
    def k11_count():
        eng.maf_runs_cigar_text(n, nruns, runs, run_off, cols, cnt=tcnt)
    ms = timed(k11_count)
    toff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    eng.exclusive_scan_u64(n, tcnt, toff)
    nbytes = int(toff[-1].item())
    text = torch.zeros(nbytes + 64, dtype=torch.uint8, device=dev)
    def k11_fill():
        eng.maf_runs_cigar_text(n, nruns, runs, run_off, cols, out=text, out_off=toff)
    ms2 = timed(lambda: (k11_count(), k11_fill())) - ms
    print("K11 runs -> cg:Z: text: count %.3f ms, fill %.3f ms  %.0f GB/s (8 B/run + %d B of text)" % (
        ms, ms2, (8 * nruns + nbytes) / ms2 / 1e6, nbytes))
