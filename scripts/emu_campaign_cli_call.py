"""`wgatools call` and `maf2paf` of the emulator build on random MAF blocks with random flags against the oracle's VCF / cg:Z: text:
python scripts/emu_campaign_cli_call.py <first seed> <seconds>.  CPU only; results: profiles/r06_emu_campaign.txt."""
import os, sys, tempfile, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import cli_cases as cc
import oracle_py as orc
from wgatools_amd import build
cli = build.CLI_EMU_BIN
t0 = time.time(); fails = 0
seed0 = int(sys.argv[1]); budget = float(sys.argv[2]); k = 0
tmp = tempfile.mkdtemp(prefix="wga_camp_")
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    rng = np.random.default_rng(s)
    nb = int(rng.integers(1, 12)); cols = int(rng.choice([1, 7, 40, 300, 1500, 4000]))
    blocks = cc._synth_maf_blocks(s, nb, cols)
    maf = os.path.join(tmp, "in.maf")
    cc._write_maf(maf, blocks)
    snp, inv = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    svlen = int(rng.choice([0, 1, 2, 5, 11, 50])); chunk = int(rng.choice([1, 3, 7, 64, 300, 5000, 1000000]))
    args = ["call", maf, "-l", str(svlen), "-c", str(chunk), "-n", "smp"] + (["-s"] if snp else []) + (["-i"] if inv else [])
    try:
        os.environ["WGA_HOST_THREADS"] = str(int(rng.choice([1, 2, 5, 64])))
        rc, out, err = cc.run(cli, *args)
        want = cc._expected_vcf(blocks, "smp", snp, inv, svlen, chunk)
        assert rc == 0 and out.decode() == want, (args, err[-300:])
        rc, out, err = cc.run(cli, "maf2paf", maf)
        lines = out.decode().splitlines()
        assert rc == 0 and len(lines) == nb, err
        for b, ln in zip(blocks, lines):
            _, txt = orc.parse_maf_seq_to_cigar(b["t"], b["q"], b["neg"])
            assert ln.split("\t")[-1] == "cg:Z:" + txt
    except Exception:
        fails += 1
        print("FAIL seed", s, args); traceback.print_exc(); sys.stdout.flush()
    if k % 20 == 0:
        print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
print("END", k, "seeds", fails, "fails")
