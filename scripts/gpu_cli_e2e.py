"""file-to-file wall time of the wgatools CLI (host parsing + PCIe + kernels + text output) on a
config-2 shaped input: N records x mean 5 kop, 2 x 50 Mb FASTA"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wgatools_amd import build, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
tmp = sys.argv[2] if len(sys.argv) > 2 else "/tmp/wga_e2e"
os.makedirs(tmp, exist_ok=True)
import torch
dev = torch.device("cuda", 0)
tb = synth.make_paf_batch_torch(9, n, 5000, 50_000_000, dev)     # records, pools and the PAF text are made in HBM
def fasta(path, name, seq):
    with open(path, "wb") as f:
        f.write(b">" + name + b"\n")
        for i in range(0, len(seq), 1 << 20):       # 1 MiB lines
            f.write(seq[i:i + (1 << 20)] + b"\n")
t_fa, q_fa, paf = os.path.join(tmp, "t.fa"), os.path.join(tmp, "q.fa"), os.path.join(tmp, "in.paf")
fasta(t_fa, b"tchr", tb["t_pool"].cpu().numpy().tobytes())
fasta(q_fa, b"qchr", tb["q_pool"].cpu().numpy().tobytes())
synth.paf_text_torch(tb).cpu().numpy().tofile(paf)
nops = tb["n_ops"]
del tb
torch.cuda.empty_cache()
print("input: %d records, %.3e ops, PAF %.1f MB" % (n, nops, os.path.getsize(paf) / 1e6))
cli = build.CLI_BIN
def run(name, args, outp, env=None):
    t0 = time.perf_counter()
    r = subprocess.run([cli] + args + ["-o", outp, "-r"], stderr=subprocess.PIPE, env=dict(os.environ, WGA_TIMING="1", **(env or {})))
    dt = time.perf_counter() - t0
    sz = os.path.getsize(outp) if os.path.exists(outp) else 0
    print("%-9s %.2f s wall  rc=%d  output %.1f MB  -> %.2e ops/s end to end" % (name, dt, r.returncode, sz / 1e6, nops / dt))
    if r.returncode: print(r.stderr.decode()[-300:])
    for ln in r.stderr.decode().splitlines():
        if ln.startswith("[timing]"): print("          " + ln)
    if os.path.isfile(outp) and not outp.endswith(".chain"): os.remove(outp)       # 15 GB of MAF, 9 GB of VCF at 100 000 records
run("stat", ["stat", "-f", "paf", paf], os.path.join(tmp, "out.tsv"))
run("paf2maf", ["paf2maf", paf, "-g", t_fa, "-q", q_fa], os.path.join(tmp, "out.maf"))
run("pafcov", ["pafcov", paf], os.path.join(tmp, "out.bed"))
run("paf2chain", ["paf2chain", paf], os.path.join(tmp, "out.chain"))
run("validate", ["validate", paf], os.path.join(tmp, "out.val"))
run("dotplot", ["dotplot", "-f", "paf", "--out-format", "csv", paf], os.path.join(tmp, "out.csv"))
run("chain2paf", ["chain2paf", os.path.join(tmp, "out.chain")], os.path.join(tmp, "out2.paf"))
run("call paf", ["call", "-f", "paf", paf, "--target", t_fa, "--query", q_fa, "-s", "-l", "50"], os.path.join(tmp, "out.vcf"))
run("pafpseudo", ["pafpseudo", paf], os.path.join(tmp, "pseudo_sym"))
all_fa = os.path.join(tmp, "all.fa")      # pafpseudo -f wants every genome in one FASTA (pseudomaf.rs:214-237)
with open(all_fa, "wb") as f:
    f.write(open(t_fa, "rb").read() + open(q_fa, "rb").read())
run("pafpseudo-f", ["pafpseudo", paf, "-f", all_fa], os.path.join(tmp, "pseudo_base"))
# the same with the csv-semantics host reader instead of the device splitter
run("stat/host", ["stat", "-f", "paf", paf], os.path.join(tmp, "out.tsv"), env={"WGA_PAF_READER": "host"})
run("p2c/host", ["paf2chain", paf], os.path.join(tmp, "out.chain"), env={"WGA_PAF_READER": "host"})
run("stat", ["stat", "-f", "paf", paf], os.path.join(tmp, "out.tsv"))
# ---- MAF commands on a config-3 shaped file: NB blocks x 1500 columns ----------------------------------
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
rng = np.random.default_rng(3)
cols = 1500
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
maf = os.path.join(tmp, "in.maf")
with open(maf, "wb") as f:
    f.write(b"##maf version=1\n")
    CH = 2000
    for c0 in range(0, nb, CH):
        m = min(CH, nb - c0)
        t = alpha[rng.integers(0, 4, (m, cols))]
        q = t.copy()
        snp = rng.random((m, cols)) < 0.012
        q[snp] = alpha[rng.integers(0, 4, int(snp.sum()))]
        gap = rng.random((m, cols)) < 0.0015
        q[gap] = 45
        gap2 = rng.random((m, cols)) < 0.0015
        t[gap2 & ~gap] = 45
        for k in range(m):
            tal = cols - int((t[k] == 45).sum()); qal = cols - int((q[k] == 45).sum())
            f.write(b"a score=255\ns\tref.chr1\t%d\t%d\t+\t250000000\t" % (1000 * (c0 + k), tal) + t[k].tobytes() +
                    b"\ns\tqry.chr1\t%d\t%d\t%s\t240000000\t" % (900 * (c0 + k), qal, b"-" if (c0 + k) % 10 == 0 else b"+") + q[k].tobytes() + b"\n\n")
ncol = nb * cols
print("MAF input: %d blocks, %.3e columns, %.1f MB" % (nb, ncol, os.path.getsize(maf) / 1e6))
def runm(name, args, outp, env=None):
    t0 = time.perf_counter()
    r = subprocess.run([cli] + args + ["-o", outp, "-r"], stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
    dt = time.perf_counter() - t0
    sz = os.path.getsize(outp) if os.path.exists(outp) else 0
    print("%-13s %.2f s wall  rc=%d  output %.1f MB  -> %.2e columns/s end to end" % (name, dt, r.returncode, sz / 1e6, ncol / dt))
    if r.returncode: print(r.stderr.decode()[-300:])
for env, tag in (({}, ""), ({"WGA_MAF_READER": "host"}, "/host")):
    runm("stat maf" + tag, ["stat", maf], os.path.join(tmp, "m.tsv"), env)
    runm("maf2paf" + tag, ["maf2paf", maf], os.path.join(tmp, "m.paf"), env)
    runm("call -s" + tag, ["call", "-s", "-l", "50", maf], os.path.join(tmp, "m.vcf"), env)
    runm("maf2chain" + tag, ["maf2chain", maf], os.path.join(tmp, "m.chain"), env)
