"""paf2maf file to file at configs[1] size against the number of pwrite threads of the copy-out (WGA_WRITE_THREADS)"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wgatools_amd import build, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tmp = "/tmp/wga_wt"; os.makedirs(tmp, exist_ok=True)
dev = torch.device("cuda", 0)
tb = synth.make_paf_batch_torch(9, n, 5000, 50_000_000, dev)
for path, name, pool in ((tmp + "/t.fa", b"tchr", tb["t_pool"]), (tmp + "/q.fa", b"qchr", tb["q_pool"])):
    seq = pool.cpu().numpy().tobytes()
    with open(path, "wb") as f:
        f.write(b">" + name + b"\n")
        for i in range(0, len(seq), 1 << 20): f.write(seq[i:i + (1 << 20)] + b"\n")
synth.paf_text_torch(tb).cpu().numpy().tofile(tmp + "/in.paf")
del tb; torch.cuda.empty_cache()
for thr in sys.argv[2:] or ["8", "16", "24", "32", "8", "16"]:
    outp = tmp + "/o.maf"
    t0 = time.perf_counter()
    r = subprocess.run([build.CLI_BIN, "paf2maf", tmp + "/in.paf", "-g", tmp + "/t.fa", "-q", tmp + "/q.fa", "-o", outp, "-r"],
                       stderr=subprocess.PIPE, env=dict(os.environ, WGA_TIMING="1", WGA_WRITE_THREADS=thr))
    dt = time.perf_counter() - t0
    ph = [l for l in r.stderr.decode().splitlines() if l.startswith("[timing]")]
    print("threads %2s: %.2f s wall rc=%d  %s" % (thr, dt, r.returncode, ph[-1][ph[-1].index("kernels"):] if ph else ""), flush=True)
    os.remove(outp)
