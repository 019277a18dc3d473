"""K19 (wga_maf_call_vcf) on n blocks x 1 500 columns against the oracle, EVERY block byte for byte (the first differences printed)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from wgatools_amd import engine
import oracle_py as orc
import test_gpu_parity as tg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
svlen = int(sys.argv[2]) if len(sys.argv) > 2 else 2
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dev = torch.device("cuda", 0)
eng = engine.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
L = 1500
t, q, n = tg._synthetic_maf_rows(dev, n, L, 23)
rows = torch.cat([t, q]); tot = n * L
cols = torch.full((n,), L, dtype=torch.int64, device=dev)
t_off = torch.arange(n, device=dev, dtype=torch.int64) * L; q_off = t_off + tot
crun = torch.zeros(n, dtype=torch.int64, device=dev)
eng.maf_call_runs(n, rows, t_off, q_off, cols, run_cnt=crun)
off = torch.zeros(n + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(crun, 0)
nrun = int(off[-1])
runs = torch.zeros(3 * nrun + 3, dtype=torch.int64, device=dev)
eng.maf_call_runs(n, rows, t_off, q_off, cols, run_cnt=crun, runs=runs, run_off=off)
recs = np.zeros(n, dtype=engine.MAF_VCF_REC_DTYPE)
recs["t_name_off"], recs["t_name_len"], recs["q_name_off"], recs["q_name_len"] = 0, 8, 8, 8
recs["t_start"] = 1600 * np.arange(n, dtype=np.uint64); recs["q_start"] = 1700 * np.arange(n, dtype=np.uint64)
recs["q_size"] = 4_000_000_000; recs["q_neg"] = (np.arange(n) % 10 == 0).astype(np.uint32)
d_recs = torch.from_numpy(recs.view(np.uint8).reshape(n, -1).copy()).to(dev)
d_names = torch.tensor(list(b"ref.chr1qry.chr1\0"), dtype=torch.uint8, device=dev)
nb = torch.zeros(n, dtype=torch.int64, device=dev); err = torch.zeros((n, 2), dtype=torch.int64, device=dev)
eng.maf_call_vcf(n, rows, t_off, q_off, cols, runs, off, d_recs, d_names, True, True, svlen, chunk, nbytes=nb, err=err)
torch.cuda.synchronize()
print("errors:", int((err[:, 0] != -1).sum()))
toff = torch.zeros(n + 1, dtype=torch.int64, device=dev); toff[1:] = torch.cumsum(nb, 0)
ntext = int(toff[-1])
text = torch.zeros(ntext + 64, dtype=torch.uint8, device=dev)
eng.maf_call_vcf(n, rows, t_off, q_off, cols, runs, off, d_recs, d_names, True, True, svlen, chunk, out=text, out_off=toff)
torch.cuda.synchronize()
host = text[:ntext].cpu().numpy().tobytes()
print("rows", host.count(b"\n"), "tabs", host.count(b"\t"), "zeros", host.count(b"\0"), "bytes", ntext)
th, qh, tof = t.cpu().numpy(), q.cpu().numpy(), toff.cpu().numpy()
bad = 0
for i in range(n):
    want = orc.call_var_maf_record("ref.chr1", "qry.chr1", th[i*L:(i+1)*L].tobytes(), qh[i*L:(i+1)*L].tobytes(), int(recs["t_start"][i]), int(recs["q_start"][i]), int((qh[i*L:(i+1)*L] != 45).sum()), 4_000_000_000, bool(recs["q_neg"][i]), True, True, svlen, chunk)
    got = host[int(tof[i]):int(tof[i+1])]
    if got != want.encode():
        bad += 1
        if bad <= 3:
            print("block", i, "differs: want %d bytes, got %d" % (len(want), len(got)))
            w, g = want.encode().splitlines(), got.splitlines()
            for k in range(max(len(w), len(g))):
                a = w[k] if k < len(w) else None; b = g[k] if k < len(g) else None
                if a != b:
                    print("  row", k, "\n   want", a, "\n   got ", b); break
print("bad blocks", bad, "of", n, "(svlen %d, chunk %d)" % (svlen, chunk))
