"""file-to-file wall times of the command line at the sizes VERDICT r03 asked for: `call` (and stat / maf2paf) on a MAF of
2 000 000 blocks x 1 500 columns (configs[2]'s shape), `pafcov` over 8 targets of 100 Mb.  One JSON object on the last line
(profiles/r04_e2e_at_size.json)."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wgatools_amd import build, synth
tmp = "/tmp/wga_e2e_size"
os.makedirs(tmp, exist_ok=True)
cli = build.CLI_BIN
res = {}
def run(name, args, outp, units, unit_name, env=None):
    t0 = time.perf_counter()
    r = subprocess.run([cli] + args + ["-o", outp, "-r"], stderr=subprocess.PIPE, env=dict(os.environ, WGA_TIMING="1", **(env or {})))
    dt = time.perf_counter() - t0
    sz = os.path.getsize(outp) if os.path.exists(outp) else 0
    ph = [l for l in r.stderr.decode().splitlines() if l.startswith("[timing]")]
    res[name] = {"wall_s": round(dt, 3), "rc": r.returncode, "output_bytes": sz, unit_name + "_per_s": units / dt, "phases": ph[0] if ph else ""}
    print("%-10s %.2f s rc=%d output %.2f GB  %.3e %s/s\n    %s" % (name, dt, r.returncode, sz / 1e9, units / dt, unit_name, ph[0] if ph else r.stderr.decode()[-300:]), flush=True)
    if os.path.isfile(outp): os.remove(outp)

# ---- MAF: 2 000 000 blocks x 1 500 columns = 20 copies of 100 000 blocks with shifted coordinates -----------------------------
nb0, cols, copies = 100_000, 1500, 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
maf = os.path.join(tmp, "in.maf")
t0 = time.perf_counter()
with open(maf, "wb") as f:
    f.write(b"##maf version=1\n")
    for cp in range(copies):
        CH = 20_000
        for c0 in range(0, nb0, CH):
            t = alpha[torch.randint(0, 4, (CH, cols), device=dev, generator=g)]
            q = t.clone()
            snp = torch.rand((CH, cols), device=dev, generator=g) < 0.012
            q[snp] = alpha[torch.randint(0, 4, (int(snp.sum()),), device=dev, generator=g)]
            gq = torch.rand((CH, cols), device=dev, generator=g) < 0.0015
            q[gq] = 45
            gt = (torch.rand((CH, cols), device=dev, generator=g) < 0.0015) & ~gq
            t[gt] = 45
            tal = (cols - (t == 45).sum(1)).cpu().numpy(); qal = (cols - (q == 45).sum(1)).cpu().numpy()
            tn, qn = t.cpu().numpy(), q.cpu().numpy()
            parts = []
            for k in range(CH):
                b = cp * nb0 + c0 + k
                parts.append(b"a score=255\ns\tref.chr1\t%d\t%d\t+\t4000000000\t" % (1600 * b, tal[k]))
                parts.append(tn[k].tobytes())
                parts.append(b"\ns\tqry.chr1\t%d\t%d\t%s\t4000000000\t" % (1600 * b, qal[k], b"-" if b % 10 == 0 else b"+"))
                parts.append(qn[k].tobytes())
                parts.append(b"\n\n")
            f.write(b"".join(parts))
nb = nb0 * copies
print("MAF: %d blocks, %.2f GB, written in %.0f s" % (nb, os.path.getsize(maf) / 1e9, time.perf_counter() - t0), flush=True)
res["maf_input"] = {"blocks": nb, "columns": nb * cols, "bytes": os.path.getsize(maf)}
torch.cuda.empty_cache()
if len(sys.argv) > 1 and sys.argv[1] == "call-threads":    # how many host threads the event rules + VCF text want
    for t in (16, 32, 64, 128):
        run("call_maf_%d" % t, ["call", "-s", "-l", "50", maf], os.path.join(tmp, "m.vcf"), nb * cols, "columns", env={"WGA_HOST_THREADS": str(t)})
    os.remove(maf)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "maf-pieces":      # piece size of the MAF reader (default 1 GiB)
    for mb in (1024, 512, 256, 128):
        run("stat_maf_%dMB" % mb, ["stat", maf], os.path.join(tmp, "m.tsv"), nb * cols, "columns", env={"WGA_CHUNK_BYTES": str(mb << 20)})
        run("call_maf_%dMB" % mb, ["call", "-s", "-l", "50", maf], os.path.join(tmp, "m.vcf"), nb * cols, "columns", env={"WGA_CHUNK_BYTES": str(mb << 20)})
    os.remove(maf)
    sys.exit(0)
run("call_maf", ["call", "-s", "-l", "50", maf], os.path.join(tmp, "m.vcf"), nb * cols, "columns")
run("call_maf_again", ["call", "-s", "-l", "50", maf], os.path.join(tmp, "m.vcf"), nb * cols, "columns")
if len(sys.argv) > 1 and sys.argv[1] == "maf-only":
    run("stat_maf", ["stat", maf], os.path.join(tmp, "m.tsv"), nb * cols, "columns")
    os.remove(maf)
    print(json.dumps(res))
    sys.exit(0)
run("stat_maf", ["stat", maf], os.path.join(tmp, "m.tsv"), nb * cols, "columns")
run("maf2paf", ["maf2paf", maf], os.path.join(tmp, "m.paf"), nb * cols, "columns")
os.remove(maf)

# ---- pafcov: 8 targets of 100 Mb, 25 000 records of ~5 kop each per target -----------------------------------------------------
paf = os.path.join(tmp, "in.paf")
n_ops = 0
with open(paf, "wb") as f:
    for k in range(8):
        tb = synth.make_paf_batch_torch(100 + k, 25_000, 5000, 100_000_000, dev)
        n_ops += tb["n_ops"]
        synth.paf_text_torch(tb, t_name=b"g%02d#1#chr1" % k, q_name=b"q%02d#1#chr1" % k).cpu().numpy().tofile(f)
        del tb
        torch.cuda.empty_cache()
res["pafcov_input"] = {"targets": 8, "target_bases": 100_000_000, "records": 200_000, "ops": n_ops, "bytes": os.path.getsize(paf)}
print("PAF: %.2f GB, %.3e ops" % (os.path.getsize(paf) / 1e9, n_ops), flush=True)
run("pafcov", ["pafcov", paf], os.path.join(tmp, "cov.bed"), n_ops, "ops")
run("stat_paf", ["stat", "-f", "paf", paf], os.path.join(tmp, "p.tsv"), n_ops, "ops")
os.remove(paf)
print(json.dumps(res))
