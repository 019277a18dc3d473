"""`wgatools paf2chain`, `maf2chain`, `chain2paf` of the emulator build on random inputs against the oracle's chain / PAF text:
python scripts/emu_campaign_cli_chain.py <first seed> <seconds>.  CPU only; results: profiles/r05_emu_campaign.txt."""
import os, pathlib, shutil, sys, tempfile, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import cli_cases as cc
import parity_cases as pc
import oracle_py as orc
from wgatools_amd import build, synth
cli = build.CLI_EMU_BIN
t0 = time.time(); fails = 0
seed0 = int(sys.argv[1]); budget = float(sys.argv[2]); k = 0
tmp = pathlib.Path(tempfile.mkdtemp(prefix="wga_camp_"))
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    rng = np.random.default_rng(s)
    try:
        # paf2chain
        n = int(rng.integers(1, 30)); mean = int(rng.choice([2, 15, 80, 600]))
        b = synth.make_paf_batch(s, n, mean, 300000, use_m=bool(rng.integers(0, 2)))
        recs, want = [], []
        for i in range(n):
            cg = pc.rec_text(b, i)
            if rng.random() < 0.3:
                cg = "cg:Z:%dI%dD" % (int(rng.integers(1, 9)), int(rng.integers(1, 9))) + cg[5:] + "%dD%dI" % (int(rng.integers(1, 9)), int(rng.integers(1, 9)))
            qs, ts = int(rng.integers(0, 1000)), int(rng.integers(0, 1000))
            qe, te = qs + 10 ** 7, ts + 10 ** 7
            neg = bool(b["strand_neg"][i])
            recs.append("q%d\t%d\t%d\t%d\t%s\tt%d\t%d\t%d\t%d\t0\t0\t60\t%s" % (i % 3, 10 ** 9, qs, qe, "-" if neg else "+", i % 2, 2 * 10 ** 9, ts, te, cg))
            want.append(orc.paf2chain_record("q%d" % (i % 3), 10 ** 9, qs, qe, neg, "t%d" % (i % 2), 2 * 10 ** 9, ts, te, cg, i))
        paf = tmp / "in.paf"
        paf.write_text("\n".join(recs) + "\n")
        rc, out, err = cc.run(cli, "paf2chain", str(paf))
        assert rc == 0 and out == b"".join(want), ("paf2chain", err[-300:])
        # maf2chain
        blocks = cc._synth_maf_blocks(s, int(rng.integers(1, 10)), int(rng.choice([1, 9, 100, 900, 3000])))
        maf = str(tmp / "in.maf")
        cc._write_maf(maf, blocks)
        wantc = b"".join(orc.maf2chain_record(x["t_name"], x["t_size"], x["t_start"], x["t_align"], x["q_name"], x["q_size"], x["q_start"],
                                              x["q_align"], x["neg"], x["t"], x["q"], j) for j, x in enumerate(blocks))
        rc, out, err = cc.run(cli, "maf2chain", maf)
        assert rc == 0 and out == wantc, ("maf2chain", err[-300:])
        # chain2paf
        crecs = cc._synth_chain(s, int(rng.integers(1, 25)), max_lines=int(rng.choice([2, 10, 120, 600])))
        starts = [(int(rng.integers(0, 10 ** 6)), int(rng.integers(0, 10 ** 6))) for _ in crecs]
        ch = tmp / "in.chain"
        ch.write_text(cc._chain_text(crecs, "tchr", 10 ** 8, "qchr", 10 ** 8, starts))
        rc, out, err = cc.run(cli, "chain2paf", str(ch))
        assert rc == 0 and out == cc._expected_chain2paf(crecs, "tchr", 10 ** 8, "qchr", 10 ** 8, starts), ("chain2paf", err[-300:])
    except Exception:
        fails += 1
        print("FAIL seed", s); traceback.print_exc(); sys.stdout.flush()
    if k % 20 == 0:
        print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
shutil.rmtree(tmp, ignore_errors=True)
print("END", k, "seeds", fails, "fails")
