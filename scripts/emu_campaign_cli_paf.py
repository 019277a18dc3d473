"""`wgatools paf2maf` (plain and .gz), `pafcov` and `pafpseudo` of the emulator build on random inputs against the oracle:
python scripts/emu_campaign_cli_paf.py <first seed> <seconds>.  CPU only; results: profiles/r05_emu_campaign.txt."""
import gzip, os, pathlib, shutil, sys, tempfile, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import cli_cases as cc
import parity_cases as pc
import oracle_py as orc
from wgatools_amd import build, synth
cli = build.CLI_EMU_BIN
t0 = time.time(); fails = 0
seed0 = int(sys.argv[1]); budget = float(sys.argv[2]); k = 0
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    rng = np.random.default_rng(s)
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="wga_camp_"))
    try:
        n = int(rng.integers(1, 40)); mean = int(rng.choice([2, 15, 80, 300])); pool = int(rng.choice([4000, 60000]))
        b = synth.make_paf_batch(s, n, mean, pool, use_m=bool(rng.integers(0, 2)))
        mapq = rng.integers(0, 61, n)
        # paf2maf: plain to stdout, .gz to a file
        t_fa, q_fa, paf = cc._write_paf2maf_case(tmp, b, mapq)
        want = cc._expected_maf(b, mapq, t_fa, q_fa, n)
        rc, out, err = cc.run(cli, "paf2maf", paf, "-g", t_fa, "-q", q_fa)
        assert rc == 0 and out == want, ("paf2maf", err[-300:])
        gz = str(tmp / "o.maf.gz")
        rc, _, err = cc.run(cli, "paf2maf", paf, "-g", t_fa, "-q", q_fa, "-o", gz)
        img = open(gz, "rb").read()
        assert rc == 0 and gzip.decompress(img) == want, ("paf2maf gz", err[-300:])
        pc.bgzf_check_stream(img, want, True, one_call=False)
        # pafcov on the same file: one target
        cov = np.zeros(len(b["t_pool"]), dtype=np.uint64)
        for i in range(n):
            orc.update_cov_vec(cov, pc.rec_text(b, i), int(b["t_src_off"][i]))
        rc, out, err = cc.run(cli, "pafcov", paf)
        bed = "".join("tchr\t%d\t%d\t%d\n" % (p, p + 1, int(cov[p])) for p in range(len(cov))).encode()
        assert rc == 0 and out == bed, ("pafcov", err[-300:])
        # pafpseudo: records laid out over three targets and three queries with gaps, abutting, overlapping and contained spans
        cs = synth.class_sums(b["code"], b["length"], b["op_off"])
        tspan = (cs["mx"] + cs["d"]).astype(np.int64)
        tnames, qnames = ["tA", "tB", "t10"], ["q1", "q2", "q3"]
        tsize = {"tA": 9000, "tB": 7000, "t10": 8000}
        contigs = {q: b["q_pool"].tobytes() for q in qnames}
        for t in tnames:
            contigs[t] = pc.rand_seq(rng, tsize[t], b"ACGTacgtN")
        recs, cursor = [], {}
        for i in range(n):
            t, q = tnames[i % 3], qnames[(i // 3) % 3]
            cur = cursor.get((t, q), 0)
            mode = rng.integers(0, 4)
            start = cur + int(rng.integers(1, 60)) if mode == 0 or cur == 0 else \
                cur if mode == 1 else max(0, cur - int(rng.integers(1, 40))) if mode == 2 else max(0, cur - int(tspan[i]) - 5)
            end = start + int(tspan[i])
            if end > tsize[t]:
                continue
            cursor[(t, q)] = max(cur, end)
            qs = int(b["q_src_off"][i])
            recs.append(dict(tname=t, qname=q, tlen=tsize[t], tstart=start, tend=end, qlen=len(b["q_pool"]), qstart=qs,
                             qend=qs + int(b["q_src_len"][i]), strand="-" if b["strand_neg"][i] else "+", cg=pc.rec_text(b, i)))
        rng.shuffle(recs)
        if recs:
            ppaf = tmp / "all.paf"
            with open(ppaf, "w") as f:
                for r in recs:
                    f.write("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t0\t0\t60\t%s\n" % (r["qname"], r["qlen"], r["qstart"], r["qend"], r["strand"],
                                                                                 r["tname"], r["tlen"], r["tstart"], r["tend"], r["cg"]))
            fa = tmp / "all.fa"
            with open(fa, "wb") as f:
                for name, seq in contigs.items():
                    f.write(b">" + name.encode() + b"\n")
                    for j in range(0, len(seq), 60):
                        f.write(seq[j:j + 60] + b"\n")
            for base in (0, 1):
                outdir = tmp / ("out%d" % base)
                rc, _, err = cc.run(cli, *(["pafpseudo", str(ppaf), "-o", str(outdir)] + (["-f", str(fa)] if base else [])))
                assert rc == 0, ("pafpseudo", base, err[-300:])
                exp = cc._expected_pseudo_files(recs, contigs, base)
                assert sorted(os.listdir(outdir)) == sorted(t + ".maf" for t in exp)
                for t, text in exp.items():
                    got = open(outdir / (t + ".maf"), "rb").read()
                    assert got.split(b"\n")[:2] == text.split(b"\n")[:2] and sorted(got.split(b"\n")) == sorted(text.split(b"\n")), ("pafpseudo", base, t)
    except Exception:
        fails += 1
        print("FAIL seed", s); traceback.print_exc(); sys.stdout.flush()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if k % 10 == 0:
        print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
print("END", k, "seeds", fails, "fails")
