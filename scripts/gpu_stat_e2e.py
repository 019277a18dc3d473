"""`wgatools stat -f paf` file to file at configs[1]'s size under the reader's knobs: where do 0.5 s of "file read" go?"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgatools_amd import build, synth
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
tmp = "/tmp/wga_e2e"
os.makedirs(tmp, exist_ok=True)
dev = torch.device("cuda", 0)
tb = synth.make_paf_batch_torch(9, n, 5000, 50_000_000, dev)
paf = os.path.join(tmp, "in.paf")
synth.paf_text_torch(tb).cpu().numpy().tofile(paf)
del tb
torch.cuda.empty_cache()
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| cores:", os.cpu_count(), "| PAF %.2f GB" % (os.path.getsize(paf) / 1e9))
def run(label, env):
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        r = subprocess.run([build.CLI_BIN, "stat", "-f", "paf", paf, "-o", os.path.join(tmp, "o.tsv"), "-r"], stderr=subprocess.PIPE,
                           env=dict(os.environ, WGA_TIMING="1", **env))
        dt = time.perf_counter() - t0
        ph = [l for l in r.stderr.decode().splitlines() if l.startswith("[timing]")]
        if best is None or dt < best[0]: best = (dt, ph[0] if ph else "", r.returncode)
    print("%-34s best of 4: %.3f s rc=%d\n      %s" % (label, best[0], best[2], best[1]))
run("default", {})
run("fast exit", {"WGA_FAST_EXIT": "1"})
run("chunk 192 MB", {"WGA_CHUNK_BYTES": str(192 << 20)})
run("chunk 192 MB, fast exit", {"WGA_CHUNK_BYTES": str(192 << 20), "WGA_FAST_EXIT": "1"})
t0 = time.perf_counter(); subprocess.run([build.CLI_BIN, "--version"], stdout=subprocess.PIPE, stderr=subprocess.PIPE); print("wgatools --version: %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); subprocess.run(["/bin/true"]); print("/bin/true: %.3f s" % (time.perf_counter() - t0))
