"""throughput of the device tokeniser (K8) on config-2 shaped CIGAR text"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wgatools_amd import engine, synth

nrec, rep = 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 32
b = synth.make_paf_batch(5, nrec, 5000, 30_000_000)
lens = (b["ops"] >> 4).astype(np.int64)
codes = (b["ops"] & 15).astype(np.int64)
chars = np.array(list("MIDNSHP=XIDB"))[codes]
toks = np.char.add(lens.astype(str), chars)
off = b["op_off"].astype(np.int64)
texts = ["".join(toks[off[i]:off[i + 1]]).encode() for i in range(nrec)]
blob1 = b"".join(texts)
tl = np.array([len(t) for t in texts], dtype=np.int64)
dev = torch.device("cuda", 0)
text = torch.from_numpy(np.frombuffer(blob1, dtype=np.uint8).copy()).to(dev).repeat(rep)
text = torch.cat([text, torch.full((64,), 48, dtype=torch.uint8, device=dev)])
tlen = torch.from_numpy(np.tile(tl, rep)).to(dev)
n = nrec * rep
text_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
text_off[1:] = torch.cumsum(tlen, 0)
eng = engine.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
cnt = torch.zeros(n, dtype=torch.int64, device=dev)
err = torch.zeros(2 * n, dtype=torch.int64, device=dev)

def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    bb.record(); torch.cuda.synchronize()
    return a.elapsed_time(bb) / reps

ms_c = timed(lambda: eng.cigar_tokenise(n, text, text_off, op_cnt=cnt, err=err))
op_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
eng.exclusive_scan_u64(n, cnt, op_off)
nops = int(op_off[-1].item())
ops = torch.zeros(nops + 1, dtype=torch.int32, device=dev)
ms_f = timed(lambda: eng.cigar_tokenise(n, text, text_off, op_cnt=cnt, err=err, ops=ops, op_off=op_off))
tb = int(text_off[-1].item())
assert nops == int(b["op_off"][-1]) * rep and int(err.view(torch.int32)[0::4].abs().sum().item()) == 0
ref = torch.from_numpy(b["ops"].view(np.int32)).to(dev)
assert bool((ops[: len(ref)] == ref).all())
t0 = time.perf_counter()
for t in texts[:200]:
    eng.pack_cigar(t)
host = sum(len(t) for t in texts[:200]) / (time.perf_counter() - t0)
print("K8 tokeniser: %d records, %.2f GB text, %.3e ops (%.2f text bytes/op)" % (n, tb / 1e9, nops, tb / nops))
print("  count pass %.3f ms  %.0f GB/s of text" % (ms_c, tb / ms_c / 1e6))
print("  fill pass  %.3f ms  %.0f GB/s (text read + 4 B/op written)" % (ms_f, (tb + 4 * nops) / ms_f / 1e6))
print("  host packer (1 core, through ctypes): %.2f GB/s of text" % (host / 1e9))
