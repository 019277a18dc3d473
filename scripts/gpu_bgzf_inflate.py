"""K17: BGZF members inflated on the device against the host threads' zlib (what wgatools did until round 3):
a synthetic FASTA of N MB compressed into 64 KiB members"""
import os, sys, time, zlib, struct, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from wgatools_amd import engine
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(1)
seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), mb << 20)
seq[::61] = 10                                  # 60-column lines
text = seq.tobytes()
def member(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    p = co.compress(data) + co.flush()
    bsize = 12 + 6 + len(p) + 8
    return b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + p + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))
B = 0xFF00
t0 = time.time()
parts = [None] * ((len(text) + B - 1) // B)
def work(t, T):
    for k in range(t, len(parts), T): parts[k] = member(text[k * B:(k + 1) * B])
th = [threading.Thread(target=work, args=(t, 32)) for t in range(32)]; [x.start() for x in th]; [x.join() for x in th]
img = b"".join(parts)
print("%d MB of FASTA -> %d members, %.1f MB compressed (made in %.1f s)" % (mb, len(parts), len(img) / 1e6, time.time() - t0))
rows, p, out = [], 0, 0
for m in parts:
    rows.append((p + 18, len(m) - 26, struct.unpack_from("<I", m, len(m) - 4)[0], out)); out += rows[-1][2]; p += len(m)
tab = np.array(rows, dtype=[("a", "<u8"), ("b", "<u4"), ("c", "<u4"), ("d", "<u8")])
dev = torch.device("cuda", 0)
eng = engine.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
d_in = torch.from_numpy(np.frombuffer(img + b"\0" * 16, dtype=np.uint8).copy()).to(dev)
d_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
d_out = torch.zeros(out + 64, dtype=torch.uint8, device=dev)
d_st = torch.zeros(len(tab), dtype=torch.int32, device=dev)
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.bgzf_inflate(d_in, len(img), len(tab), d_tab, d_out, d_st); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    print("device inflate: %.2f ms = %.1f GB/s of text (%.1f GB/s of compressed input)" % (ms, out / ms / 1e6, len(img) / ms / 1e6))
assert int(d_st.abs().sum()) == 0 and d_out[:out].cpu().numpy().tobytes() == text
for T in (1, 32):
    res = [None] * len(parts)
    def inf(t):
        for k in range(t, len(parts), T): res[k] = zlib.decompress(parts[k][18:-8], -15)
    t0 = time.time(); th = [threading.Thread(target=inf, args=(t,)) for t in range(T)]; [x.start() for x in th]; [x.join() for x in th]
    dt = time.time() - t0
    print("host zlib, %2d threads: %.0f ms = %.2f GB/s of text" % (T, dt * 1e3, out / dt / 1e9))
