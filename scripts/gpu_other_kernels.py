"""throughput of the remaining op-stream kernels (K5 pafcov, K6 pafpseudo, K7 PAF call events, K9 - K12) on the
BASELINE configs[1] batch (100 000 records x mean 5 kop), or on `nrec` records of `mean_ops` ops"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wgatools_amd import engine, synth

nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
mean_ops = int(sys.argv[2]) if len(sys.argv) > 2 else 5000     # e.g. 10000 50000: the records go through the piece kernels
dev = torch.device("cuda", 0)
tb = synth.make_paf_batch_torch(0x5747415F + 2, nrec, mean_ops, 50_000_000, dev)
lib = None
if os.environ.get("WGA_LIB_VARIANT"):     # an A/B build (wgatools_amd.build.build_hip_variant)
    from wgatools_amd import _lib, build
    lib = _lib.load(os.path.join(build.ROOT, "build_variants", "libwgahip_%s.so" % os.environ["WGA_LIB_VARIANT"]))
eng = engine.Engine(0, lib)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], tb["n"], tb["n_ops"])
n, n_ops = tb["n"], tb["n_ops"]

def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

# ---- K5 pafcov: one 50 Mb target -----------------------------------------------------------------
tlen = int(tb["t_pool"].numel())
cov = torch.zeros(tlen + 1, dtype=torch.int32, device=dev)
target_id = torch.zeros(n, dtype=torch.int32, device=dev)
cov_off = torch.zeros(1, dtype=torch.int64, device=dev)
cov_len = torch.full((1,), tlen, dtype=torch.int64, device=dev)
ms_a = timed(lambda: eng.pafcov_accumulate(batch, target_id, tb["t_src_off"], cov_off, cov_len, cov, tlen))
ms_f = timed(lambda: eng.pafcov_finalize(1, cov_off, cov_len, cov))
n_meq = int(((tb["ops"] & 15) == 7).sum().item() + ((tb["ops"] & 15) == 0).sum().item())
print("K5 pafcov accumulate: %.3f ms  %.0f GB/s (4 B/op + 2 x 4 B atomics per M/= op: %.2e such ops)" % (ms_a, (4 * n_ops + 8 * n_meq) / ms_a / 1e6, n_meq))
print("K5 pafcov finalize  : %.3f ms  %.0f GB/s (8 B per target base, %d bases)" % (ms_f, 8 * tlen / ms_f / 1e6, tlen))
# ---- K6 pafpseudo -----------------------------------------------------------------------------------
cs = torch.zeros((n, 5), dtype=torch.int64, device=dev)
eng.cigar_class_sums(batch, sums=cs)
seg = (cs[:, 0] + cs[:, 2])
dst_off = torch.zeros(n, dtype=torch.int64, device=dev)
dst_off[1:] = torch.cumsum(seg, 0)[:-1]
total = int(seg.sum().item())
out = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
skip = torch.zeros(n, dtype=torch.int64, device=dev)
ref = None
k6_diag = eng.empty(n, engine.DIAG_DTYPE)           # (the wrapper would allocate one per call)
for mode, variant in ((0, 0), (0, 3), (1, 0), (1, 3)):   # one block per tile, then the streaming row kernel (the default)
    eng.set_param("pseudo_variant", variant)
    out.zero_()
    fill = lambda: eng.pafpseudo_fill(batch, mode, tb["q_pool"], int(tb["q_pool"].numel()), tb["q_src_off"], tb["q_src_len"], skip, out, dst_off, diag=k6_diag)
    ms_alone = timed(fill)                          # without the count call in front: the fill builds the class sums itself
    ts = []                                         # the protocol: wga_cigar_class_sums (the count call, not timed here), then the fill
    for _ in range(8):
        eng.cigar_class_sums(batch, sums=cs)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fill(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sum(ts[3:]) / len(ts[3:])                  # the first calls after the buffer was rewritten by another kernel run 5-15 % longer
    rd = 4 * n_ops + (int(tb["q_src_len"].sum().item()) if mode else 0)
    note = ""
    if variant == 0:
        ref = out.clone()
    if variant == 3:
        note = "; streaming row kernel, %d tiles left to the block kernel, same bytes as the block kernel: %s" % (
            eng.get_param("pseudo_stream_left_to_blocks"), bool(torch.equal(ref, out)))
        del ref
    print("K6 pafpseudo %s : %.3f ms  %.0f GB/s (4 B/op%s + %d B written; %.3f ms when the fill has to make the class sums itself%s)" % (
        "base  " if mode else "symbol", ms, (rd + total) / ms / 1e6, " + query bases" if mode else "", total, ms_alone, note))
# ---- K7 PAF call events -----------------------------------------------------------------------------
for svlen, snp in ((50, 1), (0, 0)):
    cnt = torch.zeros(n, dtype=torch.int64, device=dev)
    ms_c = timed(lambda: eng.paf_call_events(batch, svlen, snp, ev_cnt=cnt))
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    eng.exclusive_scan_u64(n, cnt, off)
    ne = int(off[-1].item())
    ev = torch.zeros(3 * ne + 3, dtype=torch.int64, device=dev)
    ms_e = timed(lambda: eng.paf_call_events(batch, svlen, snp, ev_cnt=cnt, ev=ev, ev_off=off))
    print("K7 call events svlen=%d snp=%d: count %.3f ms %.0f GB/s; fill %.3f ms %.0f GB/s (4 B/op + 24 B x %d events)" % (
        svlen, snp, ms_c, 4 * n_ops / ms_c / 1e6, ms_e, (4 * n_ops + 24 * ne) / ms_e / 1e6, ne))
print("records %d ops %.3e" % (n, n_ops))
# ---- K9 pafcov BED text ----------------------------------------------------------------------------
name = torch.tensor(list(b"g01#1#chr1"), dtype=torch.uint8, device=dev)
cnt9 = 16 << 20
covv = torch.randint(0, 200, (cnt9,), dtype=torch.int32, device=dev)
loff = torch.zeros(cnt9 + 1, dtype=torch.int64, device=dev)
ms_s = timed(lambda: eng.pafcov_format(name, covv, 30_000_000, cnt9, line_off=loff))
tot9 = int(loff[-1].item())
txt = torch.zeros(tot9 + 8, dtype=torch.uint8, device=dev)
ms_w = timed(lambda: eng.pafcov_format(name, covv, 30_000_000, cnt9, line_off=loff, out=txt))
print("K9 pafcov format: lengths+scan %.3f ms, write %.3f ms for %d lines, %.2f GB of text = %.0f GB/s (text written)" % (
    ms_s, ms_w, cnt9, tot9 / 1e9, tot9 / ms_w / 1e6))
# ---- K10 paf2chain data lines ------------------------------------------------------------------------
trim = torch.zeros((n, 4), dtype=torch.int64, device=dev)
nb10 = torch.zeros(n, dtype=torch.int64, device=dev)
dg10 = torch.zeros((n, 3), dtype=torch.int64, device=dev)
ms_c = timed(lambda: eng.cigar_chain(batch, trim=trim, nbytes=nb10, diag=dg10))
off10 = torch.zeros(n + 1, dtype=torch.int64, device=dev)
eng.exclusive_scan_u64(n, nb10, off10)
tot10 = int(off10[-1].item())
txt10 = torch.zeros(tot10 + 8, dtype=torch.uint8, device=dev)
ms_f = timed(lambda: eng.cigar_chain(batch, out=txt10, out_off=off10))
print("K10 paf2chain: count pass %.3f ms %.0f GB/s (4 B/op); fill pass %.3f ms %.0f GB/s (4 B/op + %.2f GB of text)" % (
    ms_c, 4 * n_ops / ms_c / 1e6, ms_f, (4 * n_ops + tot10) / ms_f / 1e6, tot10 / 1e9))
# ---- K12 dotplot segments ----------------------------------------------------------------------------
for cutoff in (50, 0):
    cnt12 = torch.zeros(n, dtype=torch.int64, device=dev)
    ms_c = timed(lambda: eng.cigar_dotplot(batch, cutoff, tb["t_src_off"], tb["q_src_off"], seg_cnt=cnt12))
    off12 = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    eng.exclusive_scan_u64(n, cnt12, off12)
    ns = int(off12[-1].item())
    sg = torch.zeros(5 * ns + 5, dtype=torch.int64, device=dev)
    ms_f = timed(lambda: eng.cigar_dotplot(batch, cutoff, tb["t_src_off"], tb["q_src_off"], segs=sg, seg_off=off12))
    print("K12 dotplot cutoff=%d: count %.3f ms %.0f GB/s (4 B/op); fill %.3f ms %.0f GB/s (4 B/op + 40 B x %d segments)" % (
        cutoff, ms_c, 4 * n_ops / ms_c / 1e6, ms_f, (4 * n_ops + 40 * ns) / ms_f / 1e6, ns))
# ---- K11 bridges: chain data lines (one line per 3 ops of the batch) -> ops and CIGAR text --------------
nl = min(n_ops // 3, 150_000_000)
g = torch.Generator(device=dev); g.manual_seed(5)
lines = torch.randint(0, 3000, (nl + 1, 3), dtype=torch.int64, device=dev, generator=g)
loff11 = (torch.arange(n + 1, dtype=torch.int64, device=dev) * (nl // n)).contiguous()
loff11[-1] = nl
c11 = torch.zeros(n, dtype=torch.int64, device=dev)
ms_c = timed(lambda: eng.chain_lines_ops(n, nl, lines, loff11, cnt=c11))
o11 = torch.zeros(n + 1, dtype=torch.int64, device=dev)
eng.exclusive_scan_u64(n, c11, o11)
no = int(o11[-1].item())
ops11 = torch.zeros(no + 4, dtype=torch.int32, device=dev)
ms_f = timed(lambda: eng.chain_lines_ops(n, nl, lines, loff11, out=ops11, out_off=o11))
print("K11 chain lines -> ops : count %.3f ms, fill %.3f ms %.0f GB/s (24 B/line x %d lines + 4 B x %d ops)" % (
    ms_c, ms_f, (24 * nl + 4 * no) / ms_f / 1e6, nl, no))
ms_c = timed(lambda: eng.chain_lines_cigar_text(n, nl, lines, loff11, cnt=c11))
eng.exclusive_scan_u64(n, c11, o11)
nt = int(o11[-1].item())
txt11 = torch.zeros(nt + 8, dtype=torch.uint8, device=dev)
ms_f = timed(lambda: eng.chain_lines_cigar_text(n, nl, lines, loff11, out=txt11, out_off=o11))
print("K11 chain lines -> text: count %.3f ms, fill %.3f ms %.0f GB/s (24 B/line + %.2f GB of text)" % (
    ms_c, ms_f, (24 * nl + nt) / ms_f / 1e6, nt / 1e9))
