"""`wgatools paf2maf` file to file at configs[1]'s size under the writer's knobs"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgatools_amd import build, synth
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
tmp = "/tmp/wga_e2e"
os.makedirs(tmp, exist_ok=True)
dev = torch.device("cuda", 0)
tb = synth.make_paf_batch_torch(9, n, 5000, 50_000_000, dev)
def fasta(path, name, seq):
    with open(path, "wb") as f:
        f.write(b">" + name + b"\n")
        for i in range(0, len(seq), 1 << 20):
            f.write(seq[i:i + (1 << 20)] + b"\n")
t_fa, q_fa, paf = os.path.join(tmp, "t.fa"), os.path.join(tmp, "q.fa"), os.path.join(tmp, "in.paf")
fasta(t_fa, b"tchr", tb["t_pool"].cpu().numpy().tobytes())
fasta(q_fa, b"qchr", tb["q_pool"].cpu().numpy().tobytes())
synth.paf_text_torch(tb).cpu().numpy().tofile(paf)
del tb
torch.cuda.empty_cache()
def run(label, env, reps=3):
    best = None
    outp = os.path.join(tmp, "out.maf")
    for _ in range(reps):
        if os.path.exists(outp): os.remove(outp)
        t0 = time.perf_counter()
        r = subprocess.run([build.CLI_BIN, "paf2maf", paf, "-g", t_fa, "-q", q_fa, "-o", outp, "-r"], stderr=subprocess.PIPE, env=dict(os.environ, WGA_TIMING="1", **env))
        dt = time.perf_counter() - t0
        ph = [l for l in r.stderr.decode().splitlines() if l.startswith("[timing]")]
        sz = os.path.getsize(outp)
        if best is None or dt < best[0]: best = (dt, ph[0] if ph else "", r.returncode, sz)
    print("%-40s best of %d: %.3f s rc=%d %.2f GB\n      %s" % (label, reps, best[0], best[2], best[3] / 1e9, best[1]), flush=True)
    os.remove(outp)
for t in (8, 12, 16, 24):
    run("in order, fallocate, %d writer threads" % t, {"WGA_WRITE_IN_ORDER": "1", "WGA_WRITE_THREADS": str(t)}, reps=5)
run("streams, fallocate, 16 writer threads", {"WGA_WRITE_THREADS": "16"}, reps=5)
run("in order, no fallocate, 16 writer threads", {"WGA_WRITE_IN_ORDER": "1", "WGA_WRITE_NO_FALLOCATE": "1", "WGA_WRITE_THREADS": "16"}, reps=5)
