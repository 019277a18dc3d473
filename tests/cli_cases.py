"""End-to-end drop-in cases for the `wgatools` command line, text compared byte-for-byte with
expectations built from the oracle / SURVEY Appendix B.  Imported by test_gpu_cli.py (the real
binary over libwgahip.so, on a GPU) and test_emu_cli.py (the same host code linked against the
emulator build of the kernels, on CPU); each provides the `cli` fixture."""
import os
import subprocess

import numpy as np
import pytest

import oracle_py as orc
import parity_cases as pc
from helpers import GOLDEN
from wgatools_amd import build, synth

STAT_HEADER = ("ref_name\tref_size\tref_start\tquery_name\tquery_size\tquery_start\taligned_size\t"
               "unaligned_size\tidentity\tsimilarity\tmatched\tmismatched\tins_event\tdel_event\tins_size\t"
               "del_size\tinv_event\tinv_size\tinv_ins_event\tinv_ins_size\tinv_del_event\tinv_del_size\n")


def run(cli, *args):
    r = subprocess.run([cli] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout, r.stderr.decode()


def test_stat_maf_fixture(cli):
    """Appendix B.1"""
    rc, out, err = run(cli, "stat", os.path.join(GOLDEN, "test.maf"))
    assert rc == 0, err
    assert out.decode() == STAT_HEADER + ("ref.chr8\t182411202\t181469925\tquery.chr8\t183119688\t181989421\t1000\t"
                                          "182410202\t0.99\t0.999\t990\t9\t1\t1\t8\t1\t0\t0.0\t0\t0\t0\t0\n")
    rc, out, err = run(cli, "st", "-e", os.path.join(GOLDEN, "test.maf"))
    assert out.decode().splitlines()[1].split("\t")[7] == "0"   # -e rows: unaligned_size = 0


def test_maf2paf_fixture(cli):
    """Appendix B.2"""
    rc, out, err = run(cli, "maf2paf", os.path.join(GOLDEN, "test.maf"))
    assert rc == 0, err
    assert out.decode() == ("query.chr8\t183119688\t181989421\t181990428\t+\tref.chr8\t182411202\t181469925\t"
                            "181470925\t990\t1008\t255\tNM:i:18\tcg:Z:109=1D243=1X12=1X138=1X177=1X31=1X133=8I18="
                            "1X100=2X7=1X22=\n")


def test_stat_paf_fixture(cli):
    """Appendix B.3: both records share the pair (B,300,A,300)"""
    rc, out, err = run(cli, "stat", "-f", "paf", os.path.join(GOLDEN, "testdotplot.paf"))
    assert rc == 0, err
    assert out.decode() == STAT_HEADER + "B\t300\t0\tA\t300\t0\t250\t50\t0.84\t0.84\t210\t0\t2\t2\t30\t30\t1\t50.0\t1\t10\t1\t10\n"


def test_pafcov_fixture(cli):
    """Appendix B.4"""
    rc, out, err = run(cli, "pc", os.path.join(GOLDEN, "testdotplot.paf"))
    assert rc == 0, err
    lines = out.decode().splitlines()
    assert len(lines) == 300
    cov = np.zeros(300, dtype=int)
    for a, b in ((0, 40), (60, 120), (130, 200), (200, 210), (220, 250)):
        cov[a:b] = 1
    assert lines == ["B\t%d\t%d\t%d" % (p, p + 1, cov[p]) for p in range(300)]


def _write_paf2maf_case(tmp_path, b, mapq, bad=None):
    t_fa, q_fa, paf = tmp_path / "t.fa", tmp_path / "q.fa", tmp_path / "in.paf"
    def fasta(path, name, seq):
        with open(path, "wb") as f:
            f.write(b">" + name + b" description\n")
            for i in range(0, len(seq), 70):
                f.write(seq[i:i + 70] + b"\n")
    fasta(t_fa, b"tchr", b["t_pool"].tobytes())
    fasta(q_fa, b"qchr", b["q_pool"].tobytes())
    n = len(b["strand_neg"])
    with open(paf, "w") as f:
        f.write("# synthetic\n")
        for i in range(n):
            cg = pc.rec_text(b, i) if bad is None or i != bad[0] else bad[1]
            qs, ql = int(b["q_src_off"][i]), int(b["q_src_len"][i])
            ts, tl = int(b["t_src_off"][i]), int(b["t_src_len"][i])
            f.write("qchr\t%d\t%d\t%d\t%s\ttchr\t%d\t%d\t%d\t%d\t%d\t%d\tNM:i:0\t%s\n" % (
                len(b["q_pool"]), qs, qs + ql, "-" if b["strand_neg"][i] else "+", len(b["t_pool"]), ts,
                ts + tl, 0, 0, mapq[i], cg))
    return str(t_fa), str(q_fa), str(paf)


def _expected_maf(b, mapq, t_fa, q_fa, upto):
    out = ["#maf version=1.6 convert_from=paf t_seq_path=%s q_seq_path=%s\n" % (t_fa, q_fa)]
    for i in range(upto):
        et, eq = pc.oracle_rows(b, i)
        qs, ql = int(b["q_src_off"][i]), int(b["q_src_len"][i])
        ts, tl = int(b["t_src_off"][i]), int(b["t_src_len"][i])
        neg = bool(b["strand_neg"][i])
        qstart = len(b["q_pool"]) - (qs + ql) if neg else qs
        out.append("a score=%d\ns\ttchr\t%d\t%d\t+\t%d\t%s\ns\tqchr\t%d\t%d\t%s\t%d\t%s\n\n" % (
            mapq[i], ts, tl, len(b["t_pool"]), et.decode(), qstart, ql, "-" if neg else "+",
            len(b["q_pool"]), eq.decode()))
    return "".join(out).encode()


def test_paf2maf_end_to_end(cli, tmp_path):
    b = synth.make_paf_batch(77, 60, 300, 200000)
    mapq = np.random.default_rng(1).integers(0, 61, 60)
    t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, mapq)
    outp = str(tmp_path / "out.maf")
    rc, out, err = run(cli, "paf2maf", paf, "-g", t_fa, "-q", q_fa, "-o", outp)
    assert rc == 0, err
    assert open(outp, "rb").read() == _expected_maf(b, mapq, t_fa, q_fa, 60)
    # alias + stdout + gz output
    rc, out, err = run(cli, "p2m", paf, "--target", t_fa, "--query", q_fa)
    assert rc == 0 and out == _expected_maf(b, mapq, t_fa, q_fa, 60)
    gz = str(tmp_path / "out.maf.gz")
    rc, _, err = run(cli, "p2m", paf, "-g", t_fa, "-q", q_fa, "-o", gz)
    import gzip
    assert rc == 0 and gzip.open(gz, "rb").read() == _expected_maf(b, mapq, t_fa, q_fa, 60)
    # the .gz file is BGZF: the header line as a host (zlib) member, the rows as members deflated on the device, the empty member at the end
    img = open(gz, "rb").read()
    n_members = pc.bgzf_check_stream(img, _expected_maf(b, mapq, t_fa, q_fa, 60), True, one_call=False)
    assert n_members >= 2 and len(img) < 0.45 * len(_expected_maf(b, mapq, t_fa, q_fa, 60))
    # and the engine's own readers take it back (BGZF members through the device inflate, K17): the same PAF as from the plain file
    rc, a, err = run(cli, "maf2paf", gz)
    rc2, b_, err2 = run(cli, "maf2paf", outp)
    assert rc == 0 and rc2 == 0 and a == b_ and a.count(b"\n") == 60, (err, err2)
    rc, _, err = run(cli, "p2m", paf, "-g", t_fa, "-q", q_fa, "-o", gz)        # an existing .gz is refused without -r like any output
    assert rc != 0 and "already exists" in (err if isinstance(err, str) else err.decode())


def _bgzf_write(path, data, block=0xFF00):
    """BGZF (SAM spec 4.1): gzip members of <= 64 KB with the block size in a BC extra field, then the EOF marker"""
    import struct
    import zlib
    with open(path, "wb") as f:
        for a in list(range(0, len(data), block)) + [None]:
            chunk = b"" if a is None else data[a:a + block]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            cdata = co.compress(chunk) + co.flush()
            bsize = 18 + len(cdata) + 8
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1) + cdata +
                    struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def test_paf2maf_fasta_readers(cli, tmp_path):
    """the sequence pools are built on the device from the uploaded FASTA text — several contigs,
    ragged line lengths, blank lines, CRLF, duplicate names (the first wins), plain / gzip / BGZF files — and every
    variant gives the bytes of the host faidx reader (WGA_FASTA_READER=host) and of the oracle"""
    import gzip
    b = synth.make_paf_batch(79, 40, 200, 120000)
    mapq = np.arange(40) % 61
    t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, mapq)
    want = _expected_maf(b, mapq, t_fa, q_fa, 40)
    rng = np.random.default_rng(2)

    def messy(path, name, seq, crlf):
        eol = b"\r\n" if crlf else b"\n"
        out = bytearray(b"; a comment line in front of the first header" + eol)
        out += b">decoy1 first" + eol + b"ACGTNNNN" + eol + eol
        out += b">" + name + b" the real one" + eol
        p = 0
        while p < len(seq):
            w = int(rng.integers(1, 200))
            out += seq[p:p + w] + eol
            p += w
            if rng.random() < 0.05:
                out += eol
        out += b">" + name + b" a duplicate: ignored" + eol + b"TTTTTTTT" + eol + b">decoy2" + eol + b"GG"
        open(path, "wb").write(bytes(out))
        return bytes(out)

    t2, q2 = str(tmp_path / "t2.fa"), str(tmp_path / "q2.fa")
    t_txt = messy(t2, b"tchr", b["t_pool"].tobytes(), False)
    q_txt = messy(q2, b"qchr", b["q_pool"].tobytes(), True)
    tgz, qbgz = str(tmp_path / "t2.fa.gz"), str(tmp_path / "q2.bgz.fa.gz")
    with gzip.open(tgz, "wb") as f:
        f.write(t_txt)
    _bgzf_write(qbgz, q_txt, block=3000)          # ~80 blocks: the threaded inflate path
    for tf_, qf_ in ((t_fa, q_fa), (t2, q2), (tgz, qbgz)):
        outs = []
        for reader in ("device", "host"):
            os.environ["WGA_FASTA_READER"] = reader
            try:
                rc, out, err = run(cli, "paf2maf", paf, "-g", tf_, "-q", qf_)
            finally:
                del os.environ["WGA_FASTA_READER"]
            assert rc == 0, err
            outs.append(out.split(b"\n", 1)[1])    # the header line names the FASTA paths
        assert outs[0] == outs[1] == want.split(b"\n", 1)[1], (tf_, qf_)
    # the BGZF file is inflated on the device by default (wga_bgzf_inflate); WGA_BGZF_DEVICE=0: by host threads — same bytes
    os.environ["WGA_BGZF_DEVICE"] = "0"
    try:
        rc, out, err = run(cli, "paf2maf", paf, "-g", tgz, "-q", qbgz)
    finally:
        del os.environ["WGA_BGZF_DEVICE"]
    assert rc == 0 and out.split(b"\n", 1)[1] == want.split(b"\n", 1)[1], err
    # a damaged BGZF block is an IO error, not silently truncated input
    raw = bytearray(open(qbgz, "rb").read())
    raw[len(raw) // 2] ^= 0x5A
    bad = str(tmp_path / "bad.fa.gz")
    open(bad, "wb").write(bytes(raw))
    rc, out, err = run(cli, "paf2maf", paf, "-g", t_fa, "-q", bad)
    assert rc == 1 and "ERROR" in err, err


def test_paf2maf_error_is_streamed(cli, tmp_path):
    """streaming driver: records before the failing one are written, then `ERROR <msg>`, exit 1"""
    b = synth.make_paf_batch(78, 12, 80, 30000)
    mapq = np.arange(12)
    t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, mapq, bad=(7, "cg:Z:10=3N5="))
    rc, out, err = run(cli, "paf2maf", paf, "-g", t_fa, "-q", q_fa)
    assert rc == 1 and err.strip().endswith("ERROR CIGAR OP `N` invalid")
    assert out == _expected_maf(b, mapq, t_fa, q_fa, 7)
    t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, mapq, bad=(3, "cg:Z:10=3"))
    rc, out, err = run(cli, "paf2maf", paf, "-g", t_fa, "-q", q_fa)
    assert rc == 1 and err.strip().endswith("ERROR CIGAR OP `` invalid")
    assert out == _expected_maf(b, mapq, t_fa, q_fa, 3)


def test_stat_paf_matches_oracle_aggregation(cli, tmp_path):
    """many records over a few (ref, query) pairs; f32 columns through the oracle's RecStat"""
    b = synth.make_paf_batch(79, 50, 200, 100000)
    rng = np.random.default_rng(2)
    names = [("chr%d" % rng.integers(1, 12), "q%d" % rng.integers(0, 3)) for _ in range(50)]
    paf = tmp_path / "s.paf"
    with open(paf, "w") as f:
        for i in range(50):
            f.write("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t0\t0\t60\t%s\n" % (
                names[i][1], 5000000, int(b["q_src_off"][i]), int(b["q_src_off"][i] + b["q_src_len"][i]),
                "-" if b["strand_neg"][i] else "+", names[i][0], 9000000, int(b["t_src_off"][i]),
                int(b["t_src_off"][i] + b["t_src_len"][i]), pc.rec_text(b, i)))
    rc, out, err = run(cli, "stat", "-f", "paf", str(paf))
    assert rc == 0, err
    rows = out.decode().splitlines()
    assert rows[0] + "\n" == STAT_HEADER
    # expectation: merge in first-appearance order, stable natural sort by ref name
    import collections
    groups = collections.OrderedDict()
    for i in range(50):
        c = orc.parse_paf_to_cigar(pc.rec_text(b, i), b["strand_neg"][i])
        rs = orc.recstat_from(c)
        groups.setdefault(names[i], []).append((rs, int(b["t_src_off"][i]), int(b["q_src_off"][i])))
    def natkey(s):
        import re
        return [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", s)]
    exp = []
    for (ref, q), lst in groups.items():
        tot = collections.Counter()
        inv_size = np.float32(0)
        for rs, _, _ in lst:
            for k, _t in rs._fields_:
                if k != "inv_size":
                    tot[k] += getattr(rs, k)
            inv_size = np.float32(inv_size + np.float32(rs.inv_size))
        al = tot["aligned_size"]
        ident = np.float32(tot["matched"]) / np.float32(al)
        sim = np.float32(tot["matched"] + tot["mismatched"]) / np.float32(al)
        from test_host_cli import ryu_pretty_f32
        exp.append((ref, "\t".join(str(x) for x in (
            ref, 9000000, min(min(r for _, r, _ in lst), 9000000), q, 5000000, min(min(s for _, _, s in lst), 5000000),
            al, 9000000 - al, ryu_pretty_f32(ident), ryu_pretty_f32(sim), tot["matched"], tot["mismatched"],
            tot["ins_event"], tot["del_event"], tot["ins_size"], tot["del_size"], tot["inv_event"],
            ryu_pretty_f32(inv_size), tot["inv_ins_event"], tot["inv_ins_size"], tot["inv_del_event"],
            tot["inv_del_size"]))))
    exp.sort(key=lambda t: natkey(t[0]))   # python's sort is stable
    assert rows[1:] == [e[1] for e in exp]


def _rust_bsearch_pos(keys, key):
    size, left, right = len(keys), 0, len(keys)
    while left < right:
        mid = left + size // 2
        if keys[mid] == key:
            return mid
        if keys[mid] < key:
            left = mid + 1
        else:
            right = mid
        size = right - left
    return left


def _expected_pseudo_files(recs, contigs, base):
    """pseudomaf.rs:18-210 restated in Python over the oracle's gen_pesudo_maf_by_cigar"""
    import collections
    targets = collections.OrderedDict()
    for r in recs:
        targets.setdefault(r["tname"], []).append(r)
    files = {}
    for tname, trecs in targets.items():
        queries = collections.OrderedDict()
        for r in trecs:
            lst = queries.setdefault(r["qname"], [])
            lst.insert(_rust_bsearch_pos([x["tstart"] for x in lst], r["tstart"]), r)
        out = [b"a score=0\n"]
        first = True
        target_size = 0
        for qname, lst in queries.items():
            first_query, last_end = True, 0
            for r in lst:
                target_size = r["tlen"]
                if first:
                    seq = contigs[tname][:target_size] if base else b"N" * target_size
                    out.append(b"s\t%s\t0\t%d\t+\t%d\t%s\n" % (tname.encode(), target_size, target_size, seq))
                    first = False
                if first_query:
                    out.append(b"s\t%s\t0\t%d\t+\t%d\t" % (qname.encode(), r["qlen"], r["qlen"]))
                overlap = 0
                if r["tstart"] > last_end:
                    out.append(b"-" * (r["tstart"] - last_end))
                else:
                    if last_end > r["tend"]:
                        continue
                    overlap = last_end - r["tstart"]
                last_end = r["tend"]
                q = contigs[qname][r["qstart"]:r["qend"]] if base else b""
                if base and r["strand"] == "-":
                    q = orc.reverse_complement(q)
                seg = orc.gen_pesudo_maf_by_cigar(r["cg"], q, base)
                out.append(seg[overlap:])
                first_query = False
            out.append(b"-" * (target_size - last_end))
            out.append(b"\n")
        out.append(b"\n")
        files[tname] = b"".join(out)
    return files


@pytest.mark.parametrize("base", [False, True])
def test_pafpseudo_end_to_end(cli, tmp_path, base):
    rng = np.random.default_rng(5)
    b = synth.make_paf_batch(91, 36, 60, 60000)
    cs = synth.class_sums(b["code"], b["length"], b["op_off"])
    tspan = (cs["mx"] + cs["d"]).astype(np.int64)
    tnames, qnames = ["tA", "tB", "t10"], ["q1", "q2", "q3"]
    tsize = {"tA": 9000, "tB": 7000, "t10": 8000}
    contigs = {q: b["q_pool"].tobytes() for q in qnames}
    for t in tnames:
        contigs[t] = pc.rand_seq(rng, tsize[t], b"ACGTacgtN")
    recs, cursor = [], {}
    for i in range(36):
        t, q = tnames[i % 3], qnames[(i // 3) % 3]
        cur = cursor.get((t, q), 0)
        mode = rng.integers(0, 4)   # gap / abut / partial overlap / contained
        start = cur + int(rng.integers(1, 60)) if mode == 0 or cur == 0 else \
            cur if mode == 1 else max(0, cur - int(rng.integers(1, 40))) if mode == 2 else max(0, cur - int(tspan[i]) - 5)
        end = start + int(tspan[i])
        if end > tsize[t]:
            continue
        cursor[(t, q)] = max(cur, end)
        qs = int(b["q_src_off"][i])
        recs.append(dict(tname=t, qname=q, tlen=tsize[t], tstart=start, tend=end, qlen=len(b["q_pool"]),
                         qstart=qs, qend=qs + int(b["q_src_len"][i]), strand="-" if b["strand_neg"][i] else "+",
                         cg=pc.rec_text(b, i)))
    rng.shuffle(recs)   # input order is not sorted: the sorted insertion does the work
    paf = tmp_path / "all.paf"
    with open(paf, "w") as f:
        for r in recs:
            f.write("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t0\t0\t60\t%s\n" % (
                r["qname"], r["qlen"], r["qstart"], r["qend"], r["strand"], r["tname"], r["tlen"], r["tstart"],
                r["tend"], r["cg"]))
    fa = tmp_path / "all.fa"
    with open(fa, "wb") as f:
        for name, seq in contigs.items():
            f.write(b">" + name.encode() + b"\n")
            for k in range(0, len(seq), 60):
                f.write(seq[k:k + 60] + b"\n")
    outdir = tmp_path / ("out_base" if base else "out_sym")
    args = ["pafpseudo", str(paf), "-o", str(outdir)] + (["-f", str(fa)] if base else [])
    rc, _, err = run(cli, *args)
    assert rc == 0, err
    exp = _expected_pseudo_files(recs, contigs, base)
    assert sorted(os.listdir(outdir)) == sorted(t + ".maf" for t in exp)
    for t, text in exp.items():
        got = open(outdir / (t + ".maf"), "rb").read()
        # the reference writes query rows in HashMap order: compare rows as a multiset, header first
        assert got.split(b"\n")[:2] == text.split(b"\n")[:2]
        assert sorted(got.split(b"\n")) == sorted(text.split(b"\n")), t
    # the directory exists now: refuse without -r, accept with -r, honour -g
    rc, _, err = run(cli, *args)
    assert rc == 1 and "already exists" in err
    rc, _, err = run(cli, *(args + ["-r", "-g", "tB"]))
    assert rc == 0, err


def _pseudo_line(q, ts, te, cg):
    return "%s\t1000\t0\t%d\t+\ttA\t500\t%d\t%d\t0\t0\t60\t%s\n" % (q, te - ts, ts, te, cg)


@pytest.mark.parametrize("reader", ["device", "host"])
def test_pafpseudo_errors_follow_the_walk(cli, tmp_path, reader):
    """pseudomaf.rs:147-202: only records that survive the contained / overlap logic reach
    gen_pesudo_maf_by_cigar, so a dropped record's missing tag or broken CIGAR is never seen; the first
    kept record with a problem (targets, then queries, then sorted by target start) ends the run"""
    env_before = os.environ.get("WGA_PAF_READER")
    if reader == "host":
        os.environ["WGA_PAF_READER"] = "host"
    try:
        def go(lines, name):
            paf = tmp_path / (name + ".paf")
            paf.write_text("".join(lines))
            return run(cli, "pafpseudo", str(paf), "-o", str(tmp_path / (name + "_out")))
        ok = _pseudo_line("q1", 10, 110, "cg:Z:100=")
        # contained in [10,110): dropped before its CIGAR is looked at
        rc, _, err = go([ok, _pseudo_line("q1", 20, 60, "xx:i:0"), _pseudo_line("q1", 30, 70, "cg:Z:20=5")], "dropped")
        assert rc == 0, err
        rows = open(tmp_path / "dropped_out" / "tA.maf", "rb").read().split(b"\n")
        assert rows[2] == b"s\tq1\t0\t1000\t+\t1000\t" + b"-" * 10 + orc.gen_pesudo_maf_by_cigar("cg:Z:100=", b"", False) + b"-" * 390
        # kept: without a tag / with a token the tokeniser rejects; an unknown op letter is skipped (`_ => {}`, cigar.rs:796)
        rc, _, err = go([ok, _pseudo_line("q1", 200, 240, "xx:i:0")], "notag")
        assert rc == 1 and err.strip().endswith("ERROR CIGAR start tag not found")
        rc, _, err = go([ok, _pseudo_line("q1", 200, 240, "cg:Z:20=5Q15=")], "unknown")
        assert rc == 0, err
        rows = open(tmp_path / "unknown_out" / "tA.maf", "rb").read().split(b"\n")
        assert rows[2].endswith(b"-" * 90 + orc.gen_pesudo_maf_by_cigar("cg:Z:20=5Q15=", b"", False) + b"-" * 260)
        rc, _, err = go([ok, _pseudo_line("q1", 200, 240, "cg:Z:20=M")], "nolen")
        assert rc == 1 and err.strip().endswith("ERROR CIGAR OP `=M` invalid")
        rc, _, err = go([ok, _pseudo_line("q1", 200, 240, "cg:Z:20=5")], "noop")
        assert rc == 1 and err.strip().endswith("ERROR CIGAR OP `` invalid")
        # the sorted walk meets the record at 120 (no tag) before the one at 300 (bad op), whatever the file order
        rc, _, err = go([ok, _pseudo_line("q1", 300, 340, "cg:Z:20=5"), _pseudo_line("q1", 120, 160, "xx:i:0")], "order")
        assert rc == 1 and err.strip().endswith("ERROR CIGAR start tag not found")
    finally:
        if env_before is None:
            os.environ.pop("WGA_PAF_READER", None)
        else:
            os.environ["WGA_PAF_READER"] = env_before


# ---- call (MAF) ------------------------------------------------------------------------------------
VCF_HEADER = (
    "##fileformat=VCFv4.4\n"
    '##INFO=<ID=SVLEN,Number=A,Type=Integer,Description="Length of structural variant">\n'
    '##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">\n'
    '##INFO=<ID=END,Number=1,Type=Integer,Description="End position of the longest variant described in this record">\n'
    '##INFO=<ID=INV_NEST,Number=1,Type=String,Description="Varations nested within inversion">\n'
    '##FORMAT=<ID=QI,Number=1,Type=String,Description="Query informations">\n'
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n")


def test_call_readme_golden(cli):
    """README.md:323-343 — header and every SV row byte-identical; SNP rows carry QI (caller.rs:582-593,
    the README predates it), so their first 8 columns are compared."""
    rc, out, err = run(cli, "call", os.path.join(GOLDEN, "test.maf"), "-s", "-l0")
    assert rc == 0, err
    got = out.decode().splitlines()
    golden = open(os.path.join(GOLDEN, "readme_call_test_maf_s_l0.vcf")).read().splitlines()
    assert len(got) == len(golden)
    for g, w in zip(got, golden):
        if w.startswith("#") or "SVTYPE" in w:
            assert g == w
        else:
            assert g.split("\t")[:8] == w.split("\t")[:8]
            assert g.split("\t")[8] == "GT:QI"


def _synth_maf_blocks(seed, n_blocks, cols):
    """random gapped row pairs: match/mismatch stretches, insertions, deletions, both-gap columns,
    adjacent I/D runs, lower-case and N bases, both strands"""
    rng = np.random.default_rng(seed)
    blocks = []
    for k in range(n_blocks):
        t, q = [], []
        while len(t) < cols:
            kind = rng.choice(5, p=[0.55, 0.12, 0.14, 0.14, 0.05])
            ln = int(rng.integers(1, 40 if kind == 0 else 12))
            alpha = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
            a = alpha[rng.integers(0, 9, ln)]
            if kind == 0:
                t += list(a); q += list(a)
            elif kind == 1:
                b = alpha[rng.integers(0, 9, ln)]
                t += list(a); q += list(b)
            elif kind == 2:
                t += [45] * ln; q += list(a)
            elif kind == 3:
                t += list(a); q += [45] * ln
            else:
                t += [45] * ln; q += [45] * ln
        t, q = bytes(t[:cols]), bytes(q[:cols])
        t_al, q_al = cols - t.count(b"-"), cols - q.count(b"-")
        blocks.append(dict(t_name="chrT%d" % (k % 3), t_start=int(rng.integers(0, 10000)), t_align=t_al, t_size=50000,
                           q_name="qry.%d" % (k % 2), q_start=int(rng.integers(0, 10000)), q_align=q_al,
                           q_size=40000, neg=bool(rng.integers(0, 2)), t=t, q=q))
    return blocks


def _write_maf(path, blocks, extra_sline=False):
    with open(path, "wb") as f:
        f.write(b"##maf version=1\n")
        for b in blocks:
            f.write(b"a score=255\n")
            f.write(b"s\t%s\t%d\t%d\t+\t%d\t%s\n" % (b["t_name"].encode(), b["t_start"], b["t_align"], b["t_size"], b["t"]))
            if extra_sline:
                f.write(b"s\tother.x\t5\t%d\t+\t99999\t%s\n" % (b["t_align"], b["t"]))
            f.write(b"s\t%s\t%d\t%d\t%s\t%d\t%s\n\n" % (b["q_name"].encode(), b["q_start"], b["q_align"],
                                                     b"-" if b["neg"] else b"+", b["q_size"], b["q"]))


def _expected_vcf(blocks, sample, snp, inv, svlen, chunk):
    body = "".join(orc.call_var_maf_record(b["t_name"], b["q_name"], b["t"], b["q"], b["t_start"], b["q_start"],
                                           b["q_align"], b["q_size"], b["neg"], snp, inv, svlen, chunk)
                   for b in blocks)
    return VCF_HEADER % sample + body


@pytest.mark.parametrize("snp,inv,svlen,chunk", [(True, True, 0, 1000000), (True, False, 3, 64), (False, True, 5, 7),
                                                 (True, True, 50, 300), (True, True, 1, 1)])
def test_call_synthetic_blocks(cli, tmp_path, snp, inv, svlen, chunk):
    blocks = _synth_maf_blocks(11, 7, 1500)
    maf = tmp_path / "in.maf"
    _write_maf(maf, blocks)
    args = ["call", str(maf), "-l", str(svlen), "-c", str(chunk), "-n", "smp"]
    args += ["-s"] if snp else []
    args += ["-i"] if inv else []
    rc, out, err = run(cli, *args)
    assert rc == 0, err
    assert out.decode() == _expected_vcf(blocks, "smp", snp, inv, svlen, chunk)


def test_call_and_maf2paf_on_a_long_block(cli, tmp_path):
    """a block far beyond 32 768 columns is walked piece by piece on the device (k_maf_piece_walk): `call` with its
    SV-safe chunk cuts (caller.rs:119-265) and `maf2paf`'s cg:Z: text come out as the oracle's"""
    blocks = _synth_maf_blocks(23, 1, 90000) + _synth_maf_blocks(24, 2, 700)
    maf = tmp_path / "long.maf"
    _write_maf(maf, blocks)
    for svlen, chunk in ((3, 20000), (0, 1000000), (8, 33333)):
        rc, out, err = run(cli, "call", str(maf), "-s", "-i", "-l", str(svlen), "-c", str(chunk), "-n", "smp")
        assert rc == 0, err
        assert out.decode() == _expected_vcf(blocks, "smp", True, True, svlen, chunk), (svlen, chunk)
    rc, out, err = run(cli, "maf2paf", str(maf))
    assert rc == 0, err
    lines = out.decode().splitlines()
    assert len(lines) == 3
    for b, ln in zip(blocks, lines):
        _, txt = orc.parse_maf_seq_to_cigar(b["t"], b["q"], b["neg"])
        assert ln.split("\t")[-1] == "cg:Z:" + txt


def test_call_maf_bad_base_ends_in_front_of_its_chunk(cli, tmp_path):
    """a REF / ALT character outside ACGTN is noodles-vcf's parse error (caller.rs:480-501): the rows of the blocks and of the
    block's chunks in front of the failing chunk are written (a chunk's records are collected before any is written,
    :137-141), then the command fails with the character; the rules and the rows are made on the device (wga_maf_call_vcf)"""
    blocks = _synth_maf_blocks(31, 5, 900)
    b = blocks[3]
    t, q = bytearray(b["t"]), bytearray(b["q"])
    col = next(k for k in range(450, 900) if t[k] != 45 and q[k] != 45)
    t[col], q[col] = ord("R"), ord("A")                      # an X column whose REF is no base: a SNP row's REF
    b["t"], b["q"] = bytes(t), bytes(q)
    maf = tmp_path / "in.maf"
    _write_maf(maf, blocks)
    rc, out, err = run(cli, "call", str(maf), "-s", "-i", "-l", "2", "-c", "300", "-n", "smp")
    assert rc == 1 and "invalid reference/alternate base `R`" in err, err
    # what the reference has written by then: every block in front, and this block's chunks in front of the failing one
    good = _expected_vcf(blocks[:3], "smp", True, True, 2, 300)
    assert out.decode().startswith(good)
    rest = out.decode()[len(good):]
    b_ok = dict(b)
    t2 = bytearray(b["t"]); t2[col] = ord("C" if q[col] != ord("C") else "G")
    b_ok["t"] = bytes(t2)
    full = "".join(orc.call_var_maf_record(b_ok["t_name"], b_ok["q_name"], b_ok["t"], b_ok["q"], b_ok["t_start"], b_ok["q_start"],
                                           b_ok["q_align"], b_ok["q_size"], b_ok["neg"], True, True, 2, 300))
    assert full.startswith(rest) and len(rest) < len(full)
    pos = [int(ln.split("\t")[1]) for ln in rest.splitlines()]
    t_before = b["t_start"] + sum(1 for ch in b["t"][:col] if ch != 45)
    assert all(p <= t_before + 1 for p in pos)              # nothing at or behind the bad column's chunk end is there


def test_call_maf_gz_output_in_pieces(cli, tmp_path):
    """`call -o out.vcf.gz` over a MAF read in many pieces: every piece's rows are deflated where they lie (K18) and leave on the
    helper thread while the next piece is walked; the members inflate to the plain file's bytes = the oracle's"""
    import gzip
    blocks = _synth_maf_blocks(77, 40, 1200)
    maf = tmp_path / "in.maf"
    _write_maf(maf, blocks)
    os.environ["WGA_CHUNK_BYTES"] = "20000"
    try:
        rc1, _, err1 = run(cli, "call", str(maf), "-s", "-i", "-l", "3", "-o", str(tmp_path / "plain.vcf"), "-r")
        rc2, _, err2 = run(cli, "call", str(maf), "-s", "-i", "-l", "3", "-o", str(tmp_path / "z.vcf.gz"), "-r")
    finally:
        os.environ.pop("WGA_CHUNK_BYTES", None)
    assert rc1 == 0 and rc2 == 0, (err1, err2)
    plain = open(tmp_path / "plain.vcf", "rb").read()
    assert gzip.open(tmp_path / "z.vcf.gz", "rb").read() == plain
    assert plain.decode() == _expected_vcf(blocks, "sample", True, True, 3, 1000000)


def test_call_query_selection(cli, tmp_path):
    """caller.rs:62-108: --query-name / --query-regex pick the query s-line; blocks without it are skipped"""
    blocks = _synth_maf_blocks(5, 4, 400)
    maf = tmp_path / "in.maf"
    _write_maf(maf, blocks, extra_sline=True)
    want0 = _expected_vcf([b for b in blocks if b["q_name"] == "qry.0"], "sample", True, False, 0, 1000000)
    rc, out, err = run(cli, "c", str(maf), "-s", "-l0", "--query-name", "qry.0")
    assert rc == 0, err
    assert out.decode() == want0
    rc, out, err = run(cli, "c", str(maf), "-s", "-l0", "--query-regex", r"qry\.\d")
    assert rc == 0, err
    assert out.decode() == _expected_vcf(blocks, "sample", True, False, 0, 1000000)
    rc, out, err = run(cli, "c", str(maf), "-s", "-l0", "--query-name", "absent")
    assert rc == 0 and out.decode() == VCF_HEADER % "sample"
    # Rust `regex` syntax the host's std::regex does not speak is mapped: a leading (?i), named groups, \A / \z, \x{..}
    for pat in (r"(?i)QRY\.[0-9]", r"(?P<name>qry)\.(?<d>\d)", r"\Aqry\x{2e}\d\z", r"qry\.(?:\d|zz)", r"[q][[:alpha:]]{2}\.\d+?"):
        rc, out, err = run(cli, "c", str(maf), "-s", "-l0", "--query-regex", pat)
        assert rc == 0, (pat, err)
        assert out.decode() == _expected_vcf(blocks, "sample", True, False, 0, 1000000), pat
    rc, out, err = run(cli, "c", str(maf), "-s", "-l0", "--query-regex", r"(?i)QRY\.0")
    assert rc == 0 and out.decode() == want0
    # ... and what the crate refuses at parse time is refused: look-around, back-references, an unbalanced pattern
    # ... as is the class syntax only the crate has (nested classes, set operations, negated POSIX classes)
    for pat in (r"qry(?=\.)", r"(q)\1", r"qry\.(\d", r"(?s)qry.0", r"\p{L}+", r"[q[rx]]ry\.\d", r"[a-z&&[^x]]ry\.\d", r"q[[:^digit:]]y\.\d",
                r"[a-z--[x]]ry\.\d"):
        rc, out, err = run(cli, "c", str(maf), "-s", "-l0", "--query-regex", pat)
        assert rc != 0 and "regex parse error" in (err if isinstance(err, str) else err.decode()) and out == b"", (pat, rc, err)


# ---- call (PAF) ------------------------------------------------------------------------------------
def _expected_paf_vcf(b, sample, snp, svlen, cigars=None):
    body = []
    tp, qp = b["t_pool"].tobytes(), b["q_pool"].tobytes()
    for i in range(len(b["strand_neg"])):
        qs, ql = int(b["q_src_off"][i]), int(b["q_src_len"][i])
        ts, tl = int(b["t_src_off"][i]), int(b["t_src_len"][i])
        cg = cigars[i] if cigars and cigars.get(i) else pc.rec_text(b, i)
        # paf.rs:221-237: [start, end] inclusive, clipped at the contig end, forward strand
        body.append(orc.call_within_var_paf("tchr", "qchr", cg, tp[ts:ts + tl + 1], qp[qs:qs + ql + 1], ts, ts + tl,
                                            qs, qs + ql, bool(b["strand_neg"][i]), snp, svlen))
    return (VCF_HEADER % sample + "".join(body)).encode()


@pytest.mark.parametrize("snp,svlen", [(True, 0), (False, 2), (True, 50)])
def test_call_paf_end_to_end(cli, tmp_path, snp, svlen):
    b = synth.make_paf_batch(91, 40, 250, 150000)
    t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, np.zeros(40, dtype=int))
    args = ["call", "-f", "paf", paf, "--target", t_fa, "-q", q_fa, "-l", str(svlen), "-n", "S1"]
    rc, out, err = run(cli, *(args + (["-s"] if snp else [])))
    assert rc == 0, err
    assert out == _expected_paf_vcf(b, "S1", snp, svlen)


def test_call_paf_over_host_threads(cli, tmp_path):
    b = synth.make_paf_batch(93, 30, 200, 100000)
    t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, np.zeros(30, dtype=int))
    want = _expected_paf_vcf(b, "S1", True, 2)
    for thr in ("1", "4", "30"):
        os.environ["WGA_HOST_THREADS"] = thr
        try:
            rc, out, err = run(cli, "call", "-f", "paf", paf, "--target", t_fa, "-q", q_fa, "-l", "2", "-n", "S1", "-s")
        finally:
            del os.environ["WGA_HOST_THREADS"]
        assert rc == 0 and out == want, (thr, err)


def test_call_paf_fold_errors_are_discarded(cli, tmp_path):
    """caller.rs:673,815-819: an invalid op / token ends that record's walk silently; a missing tag
    and an empty CIGAR still abort, and nothing is written (buffered driver)"""
    b = synth.make_paf_batch(92, 6, 60, 20000)
    bad = {1: "cg:Z:30=2X4I7=3N5=1X", 3: "cg:Z:12=3I5=1X4", 4: "cg:Z:9=1XX3="}
    for i, cg in bad.items():
        t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, np.zeros(6, dtype=int), bad=(i, cg))
        rc, out, err = run(cli, "c", "-f", "paf", paf, "--target", t_fa, "--query", q_fa, "-s", "-l0")
        assert rc == 0, err
        assert out == _expected_paf_vcf(b, "sample", True, 0, cigars={i: cg})
    t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, np.zeros(6, dtype=int), bad=(2, "xx:i:1"))
    rc, out, err = run(cli, "c", "-f", "paf", paf, "--target", t_fa, "--query", q_fa, "-s", "-l0")
    assert rc == 1 and out == b"" and err.strip().endswith("ERROR CIGAR start tag not found")
    rc, out, err = run(cli, "c", "-f", "paf", paf)
    assert rc == 1 and err.strip().endswith("ERROR target and query are necessary")


def test_call_paf_rows_from_the_device(cli, tmp_path):
    """the VCF rows are formatted on the device (wga_paf_call_vcf): lower-case bases come out upper-case (noodles-vcf),
    several records and names, rows longer than the kernel's staging buffer (a 9 kb deletion); a base outside ACGTN and a
    REF / ALT slice beyond the fetched sequence end the run with nothing written"""
    rng = np.random.default_rng(12)
    tseq = bytes(rng.choice(list(b"acgtACGTn"), 12000).astype(np.uint8))
    qseq = bytes(rng.choice(list(b"ACGTacgt"), 3000).astype(np.uint8))
    def write(recs, tseq=tseq, qseq=qseq):
        t_fa, q_fa, paf = tmp_path / "t.fa", tmp_path / "q.fa", tmp_path / "in.paf"
        for path, name, seq in ((t_fa, b"tg#1#chr1", tseq), (q_fa, b"qry", qseq)):
            with open(path, "wb") as f:
                f.write(b">" + name + b"\n")
                for i in range(0, len(seq), 60):
                    f.write(seq[i:i + 60] + b"\n")
        with open(paf, "w") as f:
            for (q0, q1, neg, t0, t1, cg) in recs:
                f.write("qry\t%d\t%d\t%d\t%s\ttg#1#chr1\t%d\t%d\t%d\t0\t0\t60\t%s\n" % (
                    len(qseq), q0, q1, "-" if neg else "+", len(tseq), t0, t1, cg))
        return str(t_fa), str(q_fa), str(paf)
    recs = [(0, 30, False, 0, 9030, "cg:Z:5=1X4=9000D3=3X2=4I8="),
            (100, 1100, True, 9500, 10480, "cg:Z:" + "7=1X2=2I3=1D" * 60 + "40="),
            (5, 25, False, 11000, 11020, "cg:Z:20="),
            (2000, 2100, True, 200, 320, "cg:Z:50=20D10=1X39=")]
    t_fa, q_fa, paf = write(recs)
    for svlen, snp in ((0, True), (3, False), (10000, True)):
        rc, out, err = run(cli, *(["call", "-f", "paf", paf, "--target", t_fa, "-q", q_fa, "-l", str(svlen), "-n", "S1"]
                                  + (["-s"] if snp else [])))
        assert rc == 0, err
        want = VCF_HEADER % "S1" + "".join(
            orc.call_within_var_paf("tg#1#chr1", "qry", cg, tseq[t0:t1 + 1], qseq[q0:q1 + 1], t0, t1, q0, q1, neg, snp, svlen)
            for (q0, q1, neg, t0, t1, cg) in recs)
        assert out.decode() == want, (svlen, snp)
    bad_t = bytearray(tseq)
    bad_t[9507] = ord("R")                                   # REF of the second record's first X column
    t_fa, q_fa, paf = write(recs, tseq=bytes(bad_t))
    rc, out, err = run(cli, "call", "-f", "paf", paf, "--target", t_fa, "-q", q_fa, "-l", "0", "-s")
    assert rc == 1 and out == b"" and "invalid reference/alternate base `R`" in err, err
    t_fa, q_fa, paf = write([(0, 30, False, 11990, 12030, "cg:Z:12=1X17=")])   # the contig ends at 12 000
    rc, out, err = run(cli, "call", "-f", "paf", paf, "--target", t_fa, "-q", q_fa, "-l", "0", "-s")
    assert rc == 1 and out == b"" and "panic: VCF REF/ALT slice out of the fetched sequence" in err


def test_call_maf_index_contigs(cli, tmp_path):
    """utils.rs:414-436 + caller.rs:340-357: `<maf>.index` adds natord-sorted ##contig lines of the
    reference sequences (placement after the FORMAT lines is unpinned)"""
    import shutil
    maf = tmp_path / "t.maf"
    shutil.copy(os.path.join(GOLDEN, "test.maf"), maf)
    with open(str(maf) + ".index", "w") as f:
        f.write('{"ref.chr10":{"ivls":[{"start":1,"end":2,"strand":"+","offset":17}],"size":500,"isref":true},'
                '"query.chr8":{"ivls":[],"size":183119688,"isref":false},'
                ' "ref.chr8" : {"ivls":[], "size": 182411202, "isref": true}}')
    rc, out, err = run(cli, "call", str(maf), "-l0")
    assert rc == 0, err
    lines = out.decode().splitlines()
    assert lines[7:10] == ["##contig=<ID=ref.chr8,length=182411202>", "##contig=<ID=ref.chr10,length=500>",
                           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample"]


def test_stat_paf_error_order(cli, tmp_path):
    """buffered driver: the first failing record in input order decides the message, whether it fails
    in the tokeniser (device) or in the walk (K1); nothing is written"""
    def paf(cigars):
        p = tmp_path / "e.paf"
        with open(p, "w") as f:
            for k, cg in enumerate(cigars):
                f.write("q%d\t100\t0\t10\t+\tt\t100\t0\t10\t0\t0\t0\t%s\n" % (k, cg))
        return str(p)
    cases = [
        (["cg:Z:10=", "cg:Z:5=2N3=", "cg:Z:10=", "cg:Z:10=3"], "CIGAR OP `N` invalid"),
        (["cg:Z:10=", "cg:Z:10=3", "cg:Z:5=2N3="], "CIGAR OP `` invalid"),
        (["cg:Z:10=", "cg:Z:=", "cg:Z:5=2N3="], "Parse `` Into Integer Error"),
        (["cg:Z:10=", "cg:Z:3MM", "xx:i:0"], "CIGAR OP `MM` invalid"),
        (["cg:Z:10=", "xx:i:0", "cg:Z:3MM"], "CIGAR start tag not found"),
        (["cg:Z:10=", "cg:Z:99999999999999999999M"], "Parse `99999999999999999999` Into Integer Error"),
    ]
    for cigars, msg in cases:
        rc, out, err = run(cli, "stat", "-f", "paf", paf(cigars))
        assert rc == 1 and out == b"" and err.strip().endswith("ERROR " + msg), (cigars, err)


# ---- validate ------------------------------------------------------------------
def test_validate_report_and_fix(cli, tmp_path):
    """validate.rs:44-141: expected ends from the CIGAR (inv_* slots for strand '-'), report text of
    Display + writeln (trailing blank line), --fix rows = csv serialisation of the corrected records"""
    paf = tmp_path / "v.paf"
    rows = ["q1\t100\t10\t30\t+\tt1\t200\t5\t25\t0\t0\t60\tcg:Z:10=2I8=\tNM:i:2",     # target end should be 23
            "q2\t100\t0\t20\t-\tt1\t200\t0\t20\t0\t0\t60\tcg:Z:10=3D7=",               # query end should be 17
            "q3\t50\t0\t12\t+\tt2\t60\t1\t13\t0\t0\t0\tcg:Z:5=1X6=",                    # fine
            "q4\t50\t3\t9\t+\tt2\t60\t1\t9\t0\t0\t0\tcs:Z::4*ag:2"]                       # cs tag: 4M1X2M -> q 10, t 8
    paf.write_text("\n".join(rows) + "\n")
    rc, out, err = run(cli, "validate", str(paf))
    assert rc == 0, err
    assert out.decode() == ("Total records: 4\nQuery invalid records: 2\nTarget invalid records: 2\n"
                            "Query invalid list:\nq2:0-20\nq4:3-9\nTarget invalid list:\nt1:5-25\nt2:1-9\n\n")
    fixed = tmp_path / "fixed.paf"
    rc, out2, err = run(cli, "vf", str(paf), "-f", str(fixed), "-o", str(tmp_path / "rep.txt"))
    assert rc == 0, err
    assert (tmp_path / "rep.txt").read_bytes() == out
    want = [rows[0].replace("\t5\t25\t", "\t5\t23\t"), rows[1].replace("\t0\t20\t-", "\t0\t17\t-"), rows[2],
            rows[3].replace("\t3\t9\t+", "\t3\t10\t+").replace("\t1\t9\t0", "\t1\t8\t0")]
    assert fixed.read_text() == "\n".join(want) + "\n"
    rc, out, err = run(cli, "validate", str(paf), "--fix", str(paf))
    assert rc == 1 and err.strip().endswith("ERROR fixed file should not be the same as output file")
    paf.write_text(rows[0] + "\n" + "q9\t9\t0\t5\t+\tt\t9\t0\t5\t0\t0\t0\tcg:Z:3=2N\n")
    rc, out, err = run(cli, "validate", str(paf))
    assert rc == 1 and out == b"" and "CIGAR OP `N` invalid" in err


# ---- paf2chain ---------------------------------------------------------------------
def test_paf2chain_end_to_end(cli, tmp_path):
    """converter.rs:148-173: header (with head / tail indel trim), data lines, blank line per record, chain id =
    record index; a failing record writes nothing of itself and ends the run (streaming driver)"""
    b = synth.make_paf_batch(63, 40, 400, 300000)
    rng = np.random.default_rng(4)
    n = len(b["strand_neg"])
    recs, want = [], []
    for i in range(n):
        cg = pc.rec_text(b, i)
        if i % 5 == 0:
            cg = "cg:Z:3I2D" + cg[5:] + "7D4I"       # head and tail indels
        qs, qe = int(rng.integers(0, 1000)), 0
        ts = int(rng.integers(0, 1000))
        # coordinates need not match the CIGAR for the converter: take generous ends
        qe, te = qs + 10 ** 7, ts + 10 ** 7
        neg = bool(b["strand_neg"][i])
        recs.append("q%d\t%d\t%d\t%d\t%s\tt%d\t%d\t%d\t%d\t0\t0\t60\t%s" % (i % 3, 10 ** 9, qs, qe, "-" if neg else "+", i % 2,
                                                                  2 * 10 ** 9, ts, te, cg))
        want.append(orc.paf2chain_record("q%d" % (i % 3), 10 ** 9, qs, qe, neg, "t%d" % (i % 2), 2 * 10 ** 9, ts, te, cg, i))
    paf = tmp_path / "in.paf"
    paf.write_text("\n".join(recs) + "\n")
    rc, out, err = run(cli, "paf2chain", str(paf))
    assert rc == 0, err
    assert out == b"".join(want)
    bad = list(recs)
    bad[7] = bad[7].rsplit("\t", 1)[0] + "\tcg:Z:10=3N5="
    paf.write_text("\n".join(bad) + "\n")
    rc, out, err = run(cli, "p2c", str(paf))
    assert rc == 1 and err.strip().endswith("ERROR CIGAR OP `N` invalid")
    assert out == b"".join(want[:7])


# ---- the other chain converters --------------------------------------------------------
def test_maf2chain_end_to_end(cli, tmp_path):
    """converter.rs:57-91: header from the s-lines (strand-aware query coordinates, trims), data lines over
    cigar_cat groups, chain id = block index"""
    blocks = _synth_maf_blocks(91, 14, 900)
    # blocks that start / end with indels, and one without any aligned column
    blocks[3]["t"], blocks[3]["q"] = b"--AC" + blocks[3]["t"] + b"GT---", b"TTA-" + blocks[3]["q"] + b"--ACG"
    blocks[5]["t"], blocks[5]["q"] = b"ACGT----", b"----ACGT"
    maf = str(tmp_path / "in.maf")
    _write_maf(maf, blocks)
    want = b"".join(orc.maf2chain_record(b["t_name"], b["t_size"], b["t_start"], b["t_align"], b["q_name"], b["q_size"],
                                         b["q_start"], b["q_align"], b["neg"], b["t"], b["q"], k)
                    for k, b in enumerate(blocks))
    rc, out, err = run(cli, "maf2chain", maf)
    assert rc == 0, err
    assert out == want
    # -q picks another row of the block; an unknown name fails at the first block
    _write_maf(maf, blocks, extra_sline=True)
    rc, out, err = run(cli, "m2c", maf, "-q", blocks[0]["q_name"])
    first = [k for k, b in enumerate(blocks) if b["q_name"] != blocks[0]["q_name"]][0]
    assert rc == 1 and ("Query name:%s not found in MAF" % blocks[0]["q_name"]) in err
    assert out == b"".join(orc.maf2chain_record(b["t_name"], b["t_size"], b["t_start"], b["t_align"], b["q_name"],
                                                b["q_size"], b["q_start"], b["q_align"], b["neg"], b["t"], b["q"], k)
                           for k, b in enumerate(blocks[:first]))
    rc, out, err = run(cli, "m2c", maf, "-q", "other.x")
    assert rc == 0, err
    assert out == b"".join(orc.maf2chain_record(b["t_name"], b["t_size"], b["t_start"], b["t_align"], "other.x", 99999,
                                                5, b["t_align"], False, b["t"], b["t"], k)
                           for k, b in enumerate(blocks))


def _synth_chain(seed, n, max_lines=120):
    rng = np.random.default_rng(seed)
    recs = []
    for k in range(n):
        nl = int(rng.integers(1, max_lines))
        lines = [(int(rng.integers(1, 60)), int(rng.integers(0, 3)) * int(rng.integers(0, 12)),
                  int(rng.integers(0, 3)) * int(rng.integers(0, 12))) for _ in range(nl)]
        lines[-1] = (lines[-1][0], 0, 0)
        t_ali = sum(s + dt for s, dt, dq in lines)
        q_ali = sum(s + dq for s, dt, dq in lines)
        recs.append(dict(lines=lines, t_ali=t_ali, q_ali=q_ali, neg=bool(rng.integers(0, 2)), id=int(rng.integers(0, 10 ** 6))))
    return recs


def _chain_text(recs, t_name, t_size, q_name, q_size, starts, eol="\n", last_full=False, blank=True):
    out = []
    for r, (ts, qs) in zip(recs, starts):
        out.append("chain %d %s %d + %d %d %s %d %s %d %d %d%s" % (1000 + r["id"], t_name, t_size, ts, ts + r["t_ali"], q_name,
                                                                  q_size, "-" if r["neg"] else "+", qs, qs + r["q_ali"],
                                                                  r["id"], eol))
        for j, (s, dt, dq) in enumerate(r["lines"]):
            if j == len(r["lines"]) - 1 and not last_full:
                out.append("%d%s" % (s, eol))
            else:
                out.append("%d\t%d\t%d%s" % (s, dt, dq, eol))
        if blank:
            out.append(eol)
    return "".join(out)


def _expected_chain2paf(recs, t_name, t_size, q_name, q_size, starts):
    out = []
    for r, (ts, qs) in zip(recs, starts):
        counts, cg = orc.parse_chain_to_cigar(r["lines"], r["neg"])
        match, mism, del_bp, inv_del_bp = counts[0], counts[1], counts[5], counts[9]
        out.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t255\tcg:Z:%s\n" % (
            q_name, q_size, qs, qs + r["q_ali"], "-" if r["neg"] else "+", t_name, t_size, ts, ts + r["t_ali"], match,
            match + mism + del_bp + inv_del_bp, cg))
    return "".join(out).encode()


def test_chain2paf_end_to_end(cli, tmp_path):
    """converter.rs:391-416 + chain.rs:430-452: CIGAR '<size>M[<dq>I][<dt>D]' per data line, matches = sum of sizes,
    block length = matches + D bases; the nom reader's corner cases"""
    recs = _synth_chain(5, 30)
    starts = [(100 * k, 37 * k) for k in range(len(recs))]
    ch = tmp_path / "in.chain"
    ch.write_text(_chain_text(recs, "tchr", 10 ** 8, "qchr", 10 ** 8, starts))
    want = _expected_chain2paf(recs, "tchr", 10 ** 8, "qchr", 10 ** 8, starts)
    rc, out, err = run(cli, "chain2paf", str(ch))
    assert rc == 0, err
    assert out == want
    # \r\n line ends, three-column last lines, no blank lines between the records
    ch.write_text(_chain_text(recs, "tchr", 10 ** 8, "qchr", 10 ** 8, starts, eol="\r\n", last_full=True, blank=False), newline="")
    rc, out, err = run(cli, "c2p", str(ch))
    assert rc == 0 and out == want, err
    # ... with them, the "\r" of a blank "\r\n" line is a data line without fields for is_not("chain\n")
    ch.write_text(_chain_text(recs, "tchr", 10 ** 8, "qchr", 10 ** 8, starts, eol="\r\n"), newline="")
    rc, out, err = run(cli, "c2p", str(ch))
    assert rc == 1 and out == b"" and err.strip().endswith("ERROR Parse Chain Error By: Chain Line Field `size` Missing")
    # a last data line without newline is not a data line (line_ending fails): dropped when others precede it
    two = [dict(lines=[(5, 1, 2), (7, 0, 0)], t_ali=13, q_ali=14, neg=False, id=1)]
    ch.write_text("chain 1 t 100 + 0 13 q 100 + 0 14 1\n5\t1\t2\n7")
    cut = [dict(two[0], lines=[(5, 1, 2)])]
    rc, out, err = run(cli, "c2p", str(ch))
    assert rc == 0 and out == _expected_chain2paf(cut, "t", 100, "q", 100, [(0, 0)]).replace(b"\t0\t6\t", b"\t0\t13\t").replace(b"\t0\t7\t", b"\t0\t14\t"), (out, err)
    # ... and a record whose only data line lacks it has none: fold_many1 fails
    ch.write_text("chain 1 t 100 + 0 13 q 100 + 0 14 1\n7 and some more text")
    rc, out, err = run(cli, "c2p", str(ch))
    assert rc == 1 and out == b"" and "Format error Many1 at: 7 and some Parse Error by rust::nom, please check" in err
    # errors: nothing is written (all records are converted before the first is serialised)
    good = _chain_text(recs[:3], "t", 10 ** 8, "q", 10 ** 8, starts[:3])
    for bad, msg in (("chain 1 t 100 + 0 13 q 100 + 0 14\n5\n\n", "Parse Chain Error By: Chain Line Field `chain_id` Missing"),
                     ("chain x t 100 + 0 13 q 100 + 0 14 1\n5\n\n", "Parse `x` Into Float Error"),
                     ("chain 1e3 t 100 * 0 13 q 100 + 0 14 1\n5\n\n", "Parse Strand `*` Error"),
                     ("chain 1 t 100 + 0 13 q 100 + 0 14 1\n5\t-1\t0\n5\n\n", "Parse `-1` Into Integer Error"),
                     ("chain 1 t 100 + 0 13 q 100 + 0 14 1\n \t \n5\n\n", "Parse Chain Error By: Chain Line Field `size` Missing"),
                     ("chain 1 t 100 + 0 13 q 100 + 0 14 1\n5\n\ncomment line here\n", "Format error Tag at: comment li Parse Error by rust::nom, please check"),
                     ("track name=x and more\n", "Format error Tag at: track name Parse Error by rust::nom, please check")):
        ch.write_text(good + bad if not bad.startswith("track") else bad + good)
        rc, out, err = run(cli, "c2p", str(ch))
        assert rc == 1 and out == b"" and err.strip().endswith("ERROR " + msg), (bad, err)


def test_chain2maf_end_to_end(cli, tmp_path, to_file=False):
    """converter.rs:268-358: slices fetched on the forward strand, query reverse-complemented for '-', '-' runs
    inserted per data line (parse_chain_to_insert); score 255; the failing record ends the stream
    (to_file: through -o, the path `--gpus N` shards)"""
    def run_out(*args):
        if not to_file:
            return run(cli, *args)
        path = str(tmp_path / "out.maf")
        rc, _, err = run(cli, *args, "-o", path, "-r")
        return rc, open(path, "rb").read(), err
    rng = np.random.default_rng(8)
    recs = _synth_chain(6, 25, max_lines=200)
    T, Q = 400000, 380000
    t_pool, q_pool = pc.rand_seq(rng, T, b"ACGTacgtN"), pc.rand_seq(rng, Q, b"ACGTacgtN")
    starts = [(int(rng.integers(0, T - r["t_ali"] - 1)), int(rng.integers(0, Q - r["q_ali"] - 1))) for r in recs]
    t_fa, q_fa, ch = tmp_path / "t.fa", tmp_path / "q.fa", tmp_path / "in.chain"
    for path, name, seq in ((t_fa, b"tchr", t_pool), (q_fa, b"qchr", q_pool)):
        with open(path, "wb") as f:
            f.write(b">" + name + b"\n")
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + b"\n")
    def expected(rs, ss):
        out = ["#maf version=1.6 convert_from=chain t_seq_path=%s q_seq_path=%s\n" % (t_fa, q_fa)]
        for r, (ts, qs) in zip(rs, ss):
            t = t_pool[ts:ts + r["t_ali"]]
            q = q_pool[qs:qs + r["q_ali"]]
            if r["neg"]:
                q = orc.reverse_complement(q)
            et, eq = orc.parse_chain_to_insert(r["lines"], t, q)
            out.append("a score=255\ns\ttchr\t%d\t%d\t+\t%d\t%s\ns\tqchr\t%d\t%d\t%s\t%d\t%s\n\n" % (
                ts, r["t_ali"], T, et.decode(), Q - (qs + r["q_ali"]) if r["neg"] else qs, r["q_ali"],
                "-" if r["neg"] else "+", Q, eq.decode()))
        return "".join(out).encode()
    ch.write_text(_chain_text(recs, "tchr", T, "qchr", Q, starts))
    rc, out, err = run_out("chain2maf", str(ch), "--target", str(t_fa), "--query", str(q_fa))
    assert rc == 0, err
    assert out == expected(recs, starts)
    # record 9 claims fewer bases than its lines consume: String::insert_str panics in the reference
    broken = [dict(r) for r in recs]
    broken[9]["t_ali"] -= 40
    broken[9]["q_ali"] -= 40
    broken[9]["lines"] = broken[9]["lines"][:-1] + [(broken[9]["lines"][-1][0], 3, 0), (1, 0, 0)]
    ch.write_text(_chain_text(broken, "tchr", T, "qchr", Q, starts))
    rc, out, err = run_out("c2m", str(ch), "-g", str(t_fa), "-q", str(q_fa))
    assert rc == 1 and "panic" in err
    assert out == expected(recs[:9], starts[:9])


# ---- dotplot --out-format csv ----------------------------------------------------------
def _seg_csv(segs, t_name, q_name):
    return "".join("%d,%d,%d,%d,%s,%s,%s\n" % (s[0], s[1], s[2], s[3], "MID"[int(s[4])], t_name, q_name) for s in segs)


def test_dotplot_base_level_csv(cli, tmp_path):
    """tools/dotplot.rs base-level mode: one csv row per BasePlotdata (cigar.rs:815-985), default cutoff 50"""
    b = synth.make_paf_batch(19, 25, 500, 300000)
    n = len(b["strand_neg"])
    rng = np.random.default_rng(6)
    lines, recs = [], []
    for i in range(n):
        cg = pc.rec_text(b, i)
        if i % 4 == 0:
            cg = "cg:Z:70I3D" + cg[5:] + "120D4I2S"
        ts, qs = int(rng.integers(0, 10 ** 6)), int(rng.integers(0, 10 ** 6))
        neg = bool(b["strand_neg"][i])
        name_t, name_q = ("t,%d" % i if i == 3 else "t%d" % (i % 2)), "q%d" % (i % 3)
        lines.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t0\t0\t60\t%s" % (name_q, 10 ** 9, qs, qs + 5, "-" if neg else "+", name_t,
                                                                       10 ** 9, ts, ts + 5, cg))
        recs.append((cg, ts, qs, neg, name_t, name_q))
    paf = tmp_path / "in.paf"
    paf.write_text("\n".join(lines) + "\n")
    for args, cutoff in ((["-l", "0"], 0), ([], 50), (["--length", "7"], 7)):
        want = "ref_start,ref_end,query_start,query_end,cigar,ref_chro,query_chro\n" + "".join(
            _seg_csv(orc.cigar_to_base_plotdata(cg, ts, qs, neg, cutoff), '"%s"' % nt if "," in nt else nt, nq)
            for cg, ts, qs, neg, nt, nq in recs)
        rc, out, err = run(cli, "dotplot", "-f", "paf", "--out-format", "csv", str(paf), *args)
        assert rc == 0, err
        assert out.decode() == want
    # MAF rows: K3 runs -> ops -> the same walk; strand-aware query start (maf.rs:433-442)
    blocks = _synth_maf_blocks(33, 10, 1200)
    maf = str(tmp_path / "in.maf")
    _write_maf(maf, blocks)
    want = "ref_start,ref_end,query_start,query_end,cigar,ref_chro,query_chro\n" + "".join(
        _seg_csv(orc.maf_to_base_plotdata(k["t"], k["q"], k["t_start"], k["q_size"] - k["q_start"] - k["q_align"] if k["neg"]
                                          else k["q_start"], k["neg"], 5), k["t_name"], k["q_name"]) for k in blocks)
    rc, out, err = run(cli, "dp", maf, "--out-format", "csv", "-l", "5", "-m", "base-level")
    assert rc == 0, err
    assert out.decode() == want
    # errors leave the output empty; html needs the reference's template
    paf.write_text("\n".join(lines[:5] + [lines[5].rsplit("\t", 1)[0] + "\tcg:Z:10=3"] + lines[6:]) + "\n")
    rc, out, err = run(cli, "dotplot", "-f", "paf", "--out-format", "csv", str(paf))
    assert rc == 1 and out == b"" and "CIGAR OP `` invalid" in err
    rc, out, err = run(cli, "dotplot", "-f", "paf", str(paf))
    assert rc == 1 and "not provided by this engine" in err


def test_dotplot_overview_csv(cli, tmp_path):
    """overview mode (dotplot.rs:384-423): record extents, query pair swapped for '-', identity = matched / target span"""
    blocks = _synth_maf_blocks(34, 8, 700)
    maf = str(tmp_path / "in.maf")
    _write_maf(maf, blocks)
    rows = []
    for k in blocks:
        counts, _ = orc.parse_maf_seq_to_cigar(k["t"], k["q"], k["neg"])
        qs = k["q_size"] - k["q_start"] - k["q_align"] if k["neg"] else k["q_start"]
        qe = k["q_size"] - k["q_start"] if k["neg"] else k["q_start"] + k["q_align"]
        if k["neg"]:
            qs, qe = qe, qs
        rows.append("%d,%d,%d,%d,%s,%s,%s\n" % (k["t_start"], k["t_start"] + k["t_align"], qs, qe, repr(counts[0] / k["t_align"]),
                                               k["t_name"], k["q_name"]))
    head = "ref_start,ref_end,query_start,query_end,identity,ref_chro,query_chro\n"
    rc, out, err = run(cli, "dotplot", maf, "--out-format", "csv", "-m", "overview")
    assert rc == 0, err
    assert out.decode() == head + "".join(rows)
    rc, out, err = run(cli, "dotplot", maf, "--out-format", "csv", "-m", "overview", "-d")
    assert rc == 0 and out.decode() == head + "".join(r.rsplit(",", 3)[0] + ",1.0," + ",".join(r.rsplit(",", 2)[1:]) for r in rows), err
    # PAF: matched counts M and = (cigar.rs:629-707); an op outside M = X I D fails get_stat unless -d
    paf = tmp_path / "in.paf"
    paf.write_text("q\t1000\t10\t60\t-\tt\t2000\t100\t150\t0\t0\t60\tcg:Z:20=5X10M3I15=2D\n"
                   "q\t1000\t0\t8\t+\tt\t2000\t7\t7\t0\t0\t60\tcg:Z:8I\n")
    rc, out, err = run(cli, "dotplot", "-f", "paf", str(paf), "--out-format", "csv", "-m", "overview")
    assert rc == 0, err
    assert out.decode() == head + "100,150,60,10,0.9,t,q\n7,7,0,8,NaN,t,q\n"
    paf.write_text("q\t1000\t10\t60\t-\tt\t2000\t100\t150\t0\t0\t60\tcg:Z:20=5N\n")
    rc, out, err = run(cli, "dotplot", "-f", "paf", str(paf), "--out-format", "csv", "-m", "overview")
    assert rc == 1 and out == b"" and "CIGAR OP `N` invalid" in err
    rc, out, err = run(cli, "dotplot", "-f", "paf", str(paf), "--out-format", "csv", "-m", "overview", "-d")
    assert rc == 0 and out.decode() == head + "100,150,60,10,1.0,t,q\n"


def test_format_f64_matches_ryu_layout(cli):
    """ryu pretty::format64: plain decimals for 1e-5 <= |x| < 1e16, exponent form outside [unpinned: ryu 1.0.14]"""
    import struct
    cases = [(1.0, "1.0"), (0.1, "0.1"), (1e-5, "0.00001"), (1e-6, "1e-6"), (1.5e-7, "1.5e-7"), (1e16, "1e16"),
             (1e15, "1000000000000000.0"), (1234567890123456.0, "1234567890123456.0"), (0.3333333333333333, "0.3333333333333333"),
             (123456.789, "123456.789"), (-2.5, "-2.5"), (float("nan"), "NaN"), (float("inf"), "inf"), (1.2345e22, "1.2345e22")]
    args = ["%016x" % struct.unpack("<Q", struct.pack("<d", v))[0] for v, _ in cases]
    rc, out, err = run(cli, "__fmt_f64", *args)
    assert rc == 0 and out.decode().split() == [w for _, w in cases]


# ---- PAF input: device splitter for plain files, csv-semantics host reader otherwise ----
def test_paf_reader_selection(cli, tmp_path):
    b = synth.make_paf_batch(5, 30, 200, 100000)
    mapq = np.arange(30)
    _, _, paf = _write_paf2maf_case(tmp_path, b, mapq)
    rc, out, err = run(cli, "__paf_reader", paf)
    assert rc == 0, err
    lines = out.decode().splitlines()
    assert lines[0] == "device" and len(lines) == 31
    text = open(paf).read()
    # the same file with a quoted name, or CRLF line ends, needs the csv state machine: same records from the host reader
    for variant in (text.replace("qchr\t", '"qchr"\t', 1), text.replace("\n", "\r\n")):
        p2 = tmp_path / "v.paf"
        with open(p2, "w", newline="") as f:
            f.write(variant)
        rc, out2, err = run(cli, "__paf_reader", str(p2))
        assert rc == 0, err
        l2 = out2.decode().splitlines()
        assert l2[0] == "host" and l2[1:] == lines[1:]
    # a malformed line: the host reader reports it the way the csv crate does
    p3 = tmp_path / "bad.paf"
    p3.write_text(text + "q\t1\t2\n")
    rc, out3, err = run(cli, "stat", "-f", "paf", str(p3))
    assert rc == 1 and "CSV deserialize error" in err and "invalid length 3" in err
    # cs:Z: in place of cg:Z: is converted by the host reader (paf.rs:159-218)
    p4 = tmp_path / "cs.paf"
    p4.write_text("q\t100\t0\t10\t+\tt\t100\t0\t10\t10\t10\t60\tcs:Z::6*ag:3\n")
    rc, out4, err = run(cli, "__paf_reader", str(p4))
    assert rc == 0 and out4.decode().splitlines() == ["host", "q|100|0|10|+|t|100|0|10|10|10|60|cg:Z:6M1X3M"], (out4, err)


def test_maf_reader_selection(cli, tmp_path):
    """plain MAF files are split on the device (K14) and the rows are read where they were uploaded; the host
    reader takes over when a line needs it, with the same blocks or the reference's error"""
    blocks = _synth_maf_blocks(44, 6, 300)
    maf = str(tmp_path / "in.maf")
    _write_maf(maf, blocks, extra_sline=True)
    rc, out, err = run(cli, "__maf_reader", maf)
    assert rc == 0, err
    lines = out.decode().splitlines()
    assert lines[0] == "device" and lines[1] == "##maf version=1" and len(lines) == 2 + len(blocks)
    assert all(l.startswith("block 3 ") for l in lines[2:])
    os.environ["WGA_MAF_READER"] = "host"
    try:
        rc, out2, err = run(cli, "__maf_reader", maf)
    finally:
        del os.environ["WGA_MAF_READER"]
    assert rc == 0 and out2.decode().splitlines() == ["host"] + lines[1:], err
    # a non-ASCII byte among the fields: host reader, same blocks (U+00A0 is white space for split_whitespace only
    # in the reference; here the line simply stays with the host reader)
    text = open(maf, "rb").read()
    p2 = str(tmp_path / "v.maf")
    open(p2, "wb").write(text.replace(b"a score=255\n", b"a score=255 \xc3\xa9\n", 1))
    rc, out3, err = run(cli, "__maf_reader", p2)
    assert rc == 0 and out3.decode().splitlines()[0] == "device"          # not on an s-line: no fallback needed
    open(p2, "wb").write(text.replace(b"\tother.x\t", b"\tother.\xc3\xa9\t", 1))
    rc, out3, err = run(cli, "__maf_reader", p2)
    assert rc == 0 and out3.decode().splitlines()[0] == "host"
    # errors come from the host reader with the reference's text
    open(p2, "wb").write(text.replace(b"\t+\t99999\t", b"\t+\t", 1))
    rc, out4, err = run(cli, "stat", p2)
    # six tokens: the row text stands where the size is read, and fails to parse before `seq` is found missing
    assert rc == 1 and "Into Integer Error" in err and "Parse `" in err, err
    # maf.rs:138-211 reads the tokens left to right: a short line with a bad number reports the number
    open(p2, "wb").write(b"##maf version=1\na score=1\ns ref abc\ns q 0 1 + 10 A\n\n")
    rc, out4, err = run(cli, "stat", p2)
    assert rc == 1 and "Parse `abc` Into Integer Error" in err, err
    open(p2, "wb").write(b"##maf version=1\na score=1\ns ref 3 4 x\ns q 0 1 + 10 A\n\n")
    rc, out4, err = run(cli, "stat", p2)
    assert rc == 1 and "Parse Strand `x` Error" in err, err
    open(p2, "wb").write(b"##maf version=1\na score=1\ns ref 3\ns q 0 1 + 10 A\n\n")
    rc, out4, err = run(cli, "stat", p2)
    assert rc == 1 and "S-line Filed `align_size` Missing" in err, err
    # every MAF command gives the same bytes through both readers
    for args in (["stat"], ["maf2paf"], ["maf2chain"], ["call", "-s", "-l", "3"], ["dotplot", "--out-format", "csv", "-l", "2"]):
        rc, a, err = run(cli, *args, maf)
        assert rc == 0, (args, err)
        os.environ["WGA_MAF_READER"] = "host"
        try:
            rc, b, err = run(cli, *args, maf)
        finally:
            del os.environ["WGA_MAF_READER"]
        assert rc == 0 and a == b, args


def test_dotplot_test_html_golden(cli):
    """the reference's committed test/test.html rows (tests/golden/test_html_values.json) through the command line"""
    import json
    want = json.load(open(os.path.join(GOLDEN, "test_html_values.json")))
    rc, out, err = run(cli, "dotplot", "-f", "paf", "--out-format", "csv", "-l", "9", os.path.join(GOLDEN, "testdotplot.paf"))
    assert rc == 0, err
    rows = out.decode().splitlines()
    assert rows[0] == "ref_start,ref_end,query_start,query_end,cigar,ref_chro,query_chro"
    assert rows[1:1 + len(want)] == ["%d,%d,%d,%d,%s,%s,%s" % (w["ref_start"], w["ref_end"], w["query_start"], w["query_end"],
                                                            w["cigar"], w["ref_chro"], w["query_chro"]) for w in want]


def test_streaming_pieces_give_the_same_bytes(cli, tmp_path):
    """paf2maf, stat, validate and paf2chain read the PAF in pieces that end at line ends (1 GiB; WGA_CHUNK_BYTES for
    the test): same output whatever the piece size, chain ids and csv error positions count over the whole input"""
    b = synth.make_paf_batch(17, 50, 150, 60000)
    mapq = np.arange(50)
    t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, mapq)
    def both(*args):
        res = []
        for chunk in (None, "1500", "1"):
            if chunk:
                os.environ["WGA_CHUNK_BYTES"] = chunk
            try:
                res.append(run(cli, *args))
            finally:
                os.environ.pop("WGA_CHUNK_BYTES", None)
        return res
    for args in (["paf2maf", paf, "-g", t_fa, "-q", q_fa], ["stat", "-f", "paf", paf], ["stat", "-f", "paf", "-e", paf],
                 ["validate", paf], ["validate", paf, "-f", "-"], ["paf2chain", paf], ["pafcov", paf],
                 ["dotplot", "-f", "paf", "--out-format", "csv", "-l", "4", paf]):
        r = both(*args)
        assert r[0][0] == 0, (args, r[0][2])
        assert r[0][:2] == r[1][:2] == r[2][:2], args
    # pafcov reads a file twice (targets, then coverage) and stdin whole: same bytes
    r = subprocess.run([cli, "pafcov"], stdin=open(paf, "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == run(cli, "pafcov", paf)[1]
    # a malformed line near the end: the csv error names the same record, line and byte; paf2maf and paf2chain have
    # written everything before it
    text = open(paf).read()
    bad = str(tmp_path / "bad.paf")
    lines = text.splitlines(keepends=True)
    open(bad, "w").write("".join(lines[:40]) + "q\t1\t2\n" + "".join(lines[40:]))
    msg = lambda err: err.strip().split(" ERROR ", 1)[1]
    for args in (["stat", "-f", "paf", bad], ["paf2maf", bad, "-g", t_fa, "-q", q_fa], ["paf2chain", bad]):
        r = both(*args)
        assert r[0][0] == r[1][0] == r[2][0] == 1, args
        assert msg(r[0][2]) == msg(r[1][2]) == msg(r[2][2]) and "invalid length 3" in msg(r[0][2]), (args, r[0][2], r[1][2])
    whole, small, single = both("paf2chain", bad)
    good = run(cli, "paf2chain", paf)[1]
    assert whole[1] == b"" and good.startswith(small[1]) and good.startswith(single[1])
    assert 30 <= small[1].count(b"chain\t") <= 39          # the pieces in front of the one with the bad line
    assert single[1].count(b"chain\t") == 39                # one line per piece: every record in front of it


def test_bgzipped_paf_and_maf_inputs(cli, tmp_path):
    """a bgzipped PAF / MAF is inflated on the device (K17 behind the line reader: a run of members at a time), a plain gzip
    stream and WGA_BGZF_DEVICE=0 by zlib: the same bytes as from the plain file, whatever the piece size; a damaged member
    is an IO error"""
    import gzip
    b = synth.make_paf_batch(23, 30, 200, 50000)
    mapq = np.arange(30)
    t_fa, q_fa, paf = _write_paf2maf_case(tmp_path, b, mapq)
    text = open(paf, "rb").read()
    bgz, gz = str(tmp_path / "in.paf.bgz"), str(tmp_path / "in.paf.gz")
    _bgzf_write(bgz, text, block=700)            # ~40 members, lines across member borders
    gzip.open(gz, "wb").write(text)
    def runs(*args, **kw):
        return run(cli, *args, **kw)[:2]
    for cmd in (lambda f: ["paf2maf", f, "-g", t_fa, "-q", q_fa], lambda f: ["stat", "-f", "paf", f], lambda f: ["pafcov", f],
                lambda f: ["paf2chain", f]):
        want = runs(*cmd(paf))
        assert want[0] == 0
        assert runs(*cmd(bgz)) == want and runs(*cmd(gz)) == want, cmd("x")[0]
        for env in ({"WGA_BGZF_DEVICE": "0"}, {"WGA_CHUNK_BYTES": "900"}, {"WGA_BGZF_BATCH": "1500"},  # runs of two members each
                    {"WGA_BGZF_BATCH": "1", "WGA_CHUNK_BYTES": "900"}):
            e = dict(os.environ)
            e.update(env)
            r = subprocess.run([cli] + cmd(bgz), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
            assert (r.returncode, r.stdout) == want, (cmd("x")[0], env)
    # MAF: the paf2maf output, bgzipped
    maf_text = run(cli, "paf2maf", paf, "-g", t_fa, "-q", q_fa)[1]
    maf, mbgz = str(tmp_path / "in.maf"), str(tmp_path / "in.maf.bgz")
    open(maf, "wb").write(maf_text)
    _bgzf_write(mbgz, maf_text, block=5000)
    for cmd in (lambda f: ["stat", f], lambda f: ["maf2paf", f], lambda f: ["call", f, "-s", "-l", "3"]):
        want = runs(*cmd(maf))
        assert want[0] == 0 and runs(*cmd(mbgz)) == want, cmd("x")[0]
    # two BGZF files one behind the other (an EOF marker in the middle), and a file of the EOF marker alone
    half = len(text) // 2
    cut = text.rfind(b"\n", 0, half) + 1
    pa, pb, pab, pe = (str(tmp_path / x) for x in ("a.bgz", "b.bgz", "ab.paf.bgz", "empty.paf.bgz"))
    _bgzf_write(pa, text[:cut - 7], block=300)       # the files are cut inside a line
    _bgzf_write(pb, text[cut - 7:], block=64)
    open(pab, "wb").write(open(pa, "rb").read() + open(pb, "rb").read())
    _bgzf_write(pe, b"")
    assert runs("stat", "-f", "paf", pab) == runs("stat", "-f", "paf", paf)
    assert runs("stat", "-f", "paf", pe) == (0, b"")
    # a member whose deflate data is damaged
    raw = bytearray(open(bgz, "rb").read())
    raw[18 + 5] ^= 0xFF
    bad = str(tmp_path / "bad.paf.bgz")
    open(bad, "wb").write(bytes(raw))
    for env in ({}, {"WGA_BGZF_DEVICE": "0"}):       # the member's CRC-32 is checked on either path (gzread's message)
        r = subprocess.run([cli, "stat", "-f", "paf", bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, **env))
        assert r.returncode == 1 and r.stdout == b"" and b"IO error:" in r.stderr, r.stderr


def test_maf_streaming_pieces_give_the_same_bytes(cli, tmp_path):
    """stat, maf2paf, maf2chain and call read a MAF in pieces cut between blocks (in front of a piece's trailing run of
    s-lines); chain ids count over the whole input; a block longer than a piece is kept together"""
    blocks = _synth_maf_blocks(52, 12, 700)
    maf = str(tmp_path / "in.maf")
    _write_maf(maf, blocks, extra_sline=True)
    for args in (["stat", maf], ["stat", "-e", maf], ["maf2paf", maf], ["maf2chain", maf], ["call", "-s", "-l", "2", maf],
                 ["call", "-i", "-c", "300", maf], ["maf2paf", maf, "-q", "other.x"],
                 ["dotplot", maf, "--out-format", "csv", "-l", "3"], ["dotplot", maf, "--out-format", "csv", "-m", "overview"]):
        res = []
        for chunk in (None, "5000", "64"):
            if chunk:
                os.environ["WGA_CHUNK_BYTES"] = chunk
            try:
                res.append(run(cli, *args))
            finally:
                os.environ.pop("WGA_CHUNK_BYTES", None)
        assert res[0][0] == 0, (args, res[0][2])
        assert res[0][:2] == res[1][:2] == res[2][:2], args
        os.environ["WGA_MAF_FILL_MIN_BLOCKS"] = "1"   # the host records of a piece filled by eight threads (millions of blocks otherwise)
        try:
            assert run(cli, *args)[:2] == res[0][:2], args
        finally:
            os.environ.pop("WGA_MAF_FILL_MIN_BLOCKS", None)


# ---- ties to the reference's own fixtures (the headline functions have no reference-held vector; these two
#      round trips at least start and end on files the reference repository holds) --------------------------------------
def test_fixture_maf_paf_maf_roundtrip(cli, tmp_path):
    """test/test.maf -> maf2paf -> paf2maf gives back the fixture's two alignment rows byte for byte (parse_maf_seq_to_cigar
    cigar.rs:344-432, then parse_cigar_to_insert cigar.rs:492-551 on the block's ungapped rows as the FASTA slices).  The
    contigs are cut down to the aligned stretch (the 182 Mb around it are not in the fixture): coordinates are rebased to 0."""
    maf = open(os.path.join(GOLDEN, "test.maf"), "rb").read().decode()
    srows = [ln.split() for ln in maf.splitlines() if ln.startswith("s")]
    assert len(srows) == 2
    (_, tn, ts, tz, tstr, tsrc, trow), (_, qn, qs, qz, qstr, qsrc, qrow) = srows
    assert tstr == "+" and qstr == "+"
    rc, out, err = run(cli, "maf2paf", os.path.join(GOLDEN, "test.maf"))
    assert rc == 0, err
    f = out.decode().rstrip("\n").split("\t")
    assert f[0] == qn and f[5] == tn and int(f[2]) == int(qs) and int(f[7]) == int(ts)
    t_seq, q_seq = trow.replace("-", ""), qrow.replace("-", "")
    assert len(t_seq) == int(tz) and len(q_seq) == int(qz)
    # the same record on contigs that hold just the aligned stretch
    f[1], f[2], f[3] = str(len(q_seq)), "0", str(len(q_seq))
    f[6], f[7], f[8] = str(len(t_seq)), "0", str(len(t_seq))
    paf = tmp_path / "rt.paf"
    paf.write_text("\t".join(f) + "\n")
    t_fa, q_fa = tmp_path / "t.fa", tmp_path / "q.fa"
    t_fa.write_text(">%s\n%s\n" % (tn, t_seq))
    q_fa.write_text(">%s\n%s\n" % (qn, q_seq))
    rc, out, err = run(cli, "paf2maf", str(paf), "-g", str(t_fa), "-q", str(q_fa))
    assert rc == 0, err
    got = [ln.split("\t") for ln in out.decode().splitlines() if ln.startswith("s\t")]
    assert len(got) == 2
    assert got[0][1] == tn and got[0][3] == tz and got[0][4] == "+" and got[0][6] == trow
    assert got[1][1] == qn and got[1][3] == qz and got[1][4] == "+" and got[1][6] == qrow
    # ... and the MAF made from them reads back as the fixture's PAF record (the CIGAR of Appendix B.2)
    back = tmp_path / "back.maf"
    back.write_bytes(out)
    rc, out2, err = run(cli, "maf2paf", str(back))
    assert rc == 0, err
    assert out2.decode().rstrip("\n").split("\t")[12:] == f[12:]


def test_call_paf_readme_golden_through_maf2paf(cli, tmp_path):
    """README.md:317-343 shows ONE VCF for `call test/test.maf -s -l0` and for the same call on the PAF with the two FASTA files
    (`-f paf`).  The repository holds neither test.paf nor the FASTA files, but both follow from the fixture: `maf2paf` makes
    the record (K3's runs as cg:Z: text), the block's gap-stripped rows are the contigs' aligned stretches.  Contigs that hold
    just those stretches move every coordinate by the s lines' start fields — the golden rows are moved by the same amounts —
    and then `call -f paf` (K7 + K16: caller.rs:610-822) must print the README's rows: a reference-held vector for maf2paf's
    CIGAR and for the PAF caller (SNP rows: the first 8 columns, the README predates their QI field)"""
    maf = open(os.path.join(GOLDEN, "test.maf"), "rb").read().decode()
    (_, tn, ts, tz, _, _, trow), (_, qn, qs, qz, _, _, qrow) = [ln.split() for ln in maf.splitlines() if ln.startswith("s")]
    ts, qs = int(ts), int(qs)
    rc, out, err = run(cli, "maf2paf", os.path.join(GOLDEN, "test.maf"))
    assert rc == 0, err
    f = out.decode().rstrip("\n").split("\t")
    t_seq, q_seq = trow.replace("-", ""), qrow.replace("-", "")
    f[1], f[2], f[3] = str(len(q_seq)), "0", str(len(q_seq))
    f[6], f[7], f[8] = str(len(t_seq)), "0", str(len(t_seq))
    paf = tmp_path / "t.paf"
    paf.write_text("\t".join(f) + "\n")
    t_fa, q_fa = tmp_path / "t.fa", tmp_path / "q.fa"
    t_fa.write_text(">%s\n%s\n" % (tn, t_seq))
    q_fa.write_text(">%s\n%s\n" % (qn, q_seq))
    rc, out, err = run(cli, "call", "-f", "paf", str(paf), "--target", str(t_fa), "-q", str(q_fa), "-s", "-l0")
    assert rc == 0, err
    got = [ln for ln in out.decode().splitlines() if not ln.startswith("#")]
    golden = [ln for ln in open(os.path.join(GOLDEN, "readme_call_test_maf_s_l0.vcf")).read().splitlines() if not ln.startswith("#")]
    assert len(got) == len(golden) == 11
    for g, w in zip(got, golden):
        gc, wc = g.split("\t"), w.split("\t")
        wc[1] = str(int(wc[1]) - ts)                                      # POS on the cut-down contig
        if wc[7] != ".":
            info = dict(kv.split("=") for kv in wc[7].split(";"))
            info["END"] = str(int(info["END"]) - ts)
            wc[7] = ";".join("%s=%s" % (k, info[k]) for k in [kv.split("=")[0] for kv in w.split("\t")[7].split(";")])
            name, a, b, strand = wc[9].split(":")[1].split("@")        # 1|1:<query>@<a>@<b>@P
            wc[9] = "1|1:%s@%d@%d@%s" % (name, int(a) - qs, int(b) - qs, strand)
            assert gc == wc, (g, w)
        else:
            assert gc[:8] == wc[:8] and gc[8] == "GT:QI", (g, w)


def test_fixture_paf_chain_paf_roundtrip(cli, tmp_path):
    """test/testdotplot.paf -> paf2chain -> chain2paf: coordinates (the '+' record's; the '-' record's as the reference's header
    arithmetic leaves them), strands and the CIGAR of both records come back
    (parse_cigar_to_chain cigar.rs:251-295 with the header math of chain.rs:103-183, then parse_chain_to_cigar
    cigar.rs:554-627).  chain2paf recomputes the match / block columns from the chain (matches = sum of the block sizes,
    block length = matches + D bases, chain.rs:430-452): record 1's are the fixture's, record 2's matches column is 40 where
    the fixture says 30 — its 10M + 10M + 20M."""
    src = open(os.path.join(GOLDEN, "testdotplot.paf")).read().splitlines()
    rc, chain, err = run(cli, "paf2chain", os.path.join(GOLDEN, "testdotplot.paf"))
    assert rc == 0, err
    ch = tmp_path / "rt.chain"
    ch.write_bytes(chain)
    rc, out, err = run(cli, "chain2paf", str(ch))
    assert rc == 0, err
    got = out.decode().splitlines()
    assert len(got) == len(src) == 2
    for k, (g, w) in enumerate(zip(got, src)):
        gf, wf = g.split("\t"), w.split("\t")
        if wf[4] == "-":
            # chain.rs:168-175 (ChainHeader from a '-' strand PafRecord): query.start = size - (end - head_ins), then
            # query.end = size - (THAT start + tail_ins) — the reference's own arithmetic, which convert2paf hands back as it
            # stands (chain.rs:437-440): 300 - 250 = 50, 300 - 50 = 250
            qsz, qe = int(wf[1]), int(wf[3])
            wf[2] = str(qsz - qe)
            wf[3] = str(qsz - int(wf[2]))
        assert gf[:9] == wf[:9], (k, gf, wf)                   # names, sizes, starts, ends, strand
        assert gf[10] == wf[10]                                 # alignment block length
        assert [x for x in gf if x.startswith("cg:Z:")] == [x for x in wf if x.startswith("cg:Z:")], k
    assert got[0].split("\t")[9] == src[0].split("\t")[9] == "170" and got[1].split("\t")[9] == "40"


def pafpseudo_walk_case(cli, tmp_path, n_rec, n_targets, n_queries, check_targets):
    """BASELINE configs[4]'s all-to-all part through `pafpseudo` (symbol mode): n_rec records over n_targets x n_queries
    (target, query) pairs in shuffled input order, with gaps, abutting records, partial overlaps (the new record's first
    columns are trimmed, pseudomaf.rs:190-192) and contained records (dropped, :176-178); the grouping by target and query
    (:25-42), the sorted insertion (:86-95) and the walk (:147-210) run over all of them, and the files of `check_targets`
    targets are compared with the oracle's rows (gen_pesudo_maf_by_cigar, cigar.rs:744-804) for every record that lands
    in them."""
    from wgatools_amd.synth import make_ops, class_sums
    rng = np.random.default_rng(17)
    ops, op_off, code, length = make_ops(rng, n_rec, 12, 0.5, False)
    cs = class_sums(code, length, op_off)
    tspan = (cs["mx"] + cs["d"]).astype(np.int64)
    qspan = (cs["mx"] + cs["i"]).astype(np.int64)
    pair = rng.integers(0, n_targets * n_queries, n_rec)
    mode = rng.integers(0, 4, n_rec)
    jit = rng.integers(1, 60, n_rec)
    back = rng.integers(1, 40, n_rec)
    per_pair = n_rec // (n_targets * n_queries) + 1
    tsize = int(per_pair * (int(tspan.mean()) + 40) * 1.3) + 1000
    cursor = np.zeros(n_targets * n_queries, dtype=np.int64)
    start = np.zeros(n_rec, dtype=np.int64)
    keep = np.ones(n_rec, dtype=bool)
    for i in range(n_rec):                          # the cursor of a pair is sequential
        p_ = pair[i]
        cur = cursor[p_]
        m = mode[i]
        st = cur + jit[i] if (m == 0 or cur == 0) else cur if m == 1 else max(0, cur - back[i]) if m == 2 else max(0, cur - tspan[i] - 5)
        en = st + tspan[i]
        if en > tsize:
            keep[i] = False
            continue
        start[i] = st
        if en > cur:
            cursor[p_] = en
    order = np.flatnonzero(keep)
    rng.shuffle(order)                              # input order is not sorted: the sorted insertion does the work
    tnames = ["tg%02d#1#chr%d" % (k, 1 + k % 5) for k in range(n_targets)]
    qnames = ["qg%02d#2#ctg" % k for k in range(n_queries)]
    texts = [None] * n_rec
    paf = tmp_path / "walk.paf"
    with open(paf, "wb") as f:
        lines = []
        for i in order:
            t, q = tnames[pair[i] // n_queries], qnames[pair[i] % n_queries]
            cg = orc.ops_to_text(ops[int(op_off[i]):int(op_off[i + 1])])
            texts[i] = cg
            lines.append(b"%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t0\t0\t60\t%s\n" % (
                q.encode(), 10 ** 8, 1000 + i, 1000 + i + int(qspan[i]), b"+-"[i & 1:(i & 1) + 1], t.encode(), tsize, int(start[i]),
                int(start[i] + tspan[i]), cg))
            if len(lines) >= 50000:
                f.write(b"".join(lines))
                lines = []
        f.write(b"".join(lines))
    outdir = tmp_path / "walk_out"
    rc, _, err = run(cli, "pafpseudo", str(paf), "-o", str(outdir))
    assert rc == 0, err
    assert sorted(os.listdir(outdir)) == sorted(t + ".maf" for t in set(tnames[pair[i] // n_queries] for i in order))
    checked = 0
    for tk in range(check_targets):
        recs = [dict(tname=tnames[tk], qname=qnames[pair[i] % n_queries], tlen=tsize, tstart=int(start[i]), tend=int(start[i] + tspan[i]),
                     qlen=10 ** 8, qstart=0, qend=0, strand="+", cg=texts[i].decode()) for i in order if pair[i] // n_queries == tk]
        exp = _expected_pseudo_files(recs, {}, False)[tnames[tk]]
        got = open(outdir / (tnames[tk] + ".maf"), "rb").read()
        assert got.split(b"\n")[:2] == exp.split(b"\n")[:2]
        assert sorted(got.split(b"\n")) == sorted(exp.split(b"\n")), tnames[tk]   # query rows come in HashMap order in the reference
        checked += len(recs)
    return len(order), checked


def test_pafpseudo_walk_over_many_records(cli, tmp_path):
    n, checked = pafpseudo_walk_case(cli, tmp_path, 1200, 2, 4, 2)
    assert n > 1000 and checked == n


def test_gz_outputs_of_host_text_go_through_the_device_deflate(cli, tmp_path):
    """`-o x.gz` (utils.rs:201-209) for text the host makes: a few lines leave as a zlib member, megabytes are handed to the
    device's deflate (members of 32 768 input bytes are its mark; a host member holds up to 65 280) — either way the file is
    BGZF and inflates to what the plain output holds."""
    import gzip
    rng = np.random.default_rng(5)
    n = 9500
    maf = str(tmp_path / "many.maf")
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(maf, "wb") as f:
        f.write(b"##maf version=1\n")
        for k in range(n):
            row = alpha[rng.integers(0, 4, 24)].tobytes()
            f.write(b"a score=1\ns\tthe.target.contig.%05d\t%d\t24\t+\t900000\t%s\ns\tthe.query.contig.%05d\t%d\t24\t+\t800000\t%s\n\n"
                    % (k % 977, 7 * k, row, k % 881, 5 * k, row[:11] + b"T" + row[12:]))
    plain, gz = str(tmp_path / "many.paf"), str(tmp_path / "many.paf.gz")
    rc, _, err = run(cli, "maf2paf", maf, "-o", plain)
    assert rc == 0, err
    want = open(plain, "rb").read()
    assert len(want) > (1 << 20) and want.count(b"\n") == n
    rc, _, err = run(cli, "maf2paf", maf, "-o", gz)
    assert rc == 0, err
    img = open(gz, "rb").read()
    assert gzip.decompress(img) == want
    pc.bgzf_check_stream(img, want, True, one_call=False)
    tab, _ = pc.bgzf_table(img)
    assert (tab["out_len"] == 32768).sum() >= len(want) // 32768 - 1          # the device's members
    # a middling output (below the 1 MiB that is worth the trip to the device): zlib members of 65 280 input bytes from the host
    mid, midmaf = str(tmp_path / "mid.paf.gz"), str(tmp_path / "mid.maf")
    open(midmaf, "wb").write(b"\n\n".join(open(maf, "rb").read().split(b"\n\n")[:2500]) + b"\n\n")
    rc, want_mid, err = run(cli, "maf2paf", midmaf)
    assert rc == 0 and 3 * 65280 < len(want_mid) < (1 << 20), err
    rc, _, err = run(cli, "maf2paf", midmaf, "-o", mid)
    assert rc == 0, err
    img = open(mid, "rb").read()
    pc.bgzf_check_stream(img, want_mid, True, one_call=False)
    tab, _ = pc.bgzf_table(img)
    assert (tab["out_len"][:-2] == 65280).all() and len(tab) == len(want_mid) // 65280 + 2 and len(img) < 0.5 * len(want_mid)
    # a small output: one host member and the closing one
    small = str(tmp_path / "few.paf.gz")
    few = str(tmp_path / "few.maf")
    open(few, "wb").write(b"\n\n".join(open(maf, "rb").read().split(b"\n\n")[:3]) + b"\n\n")
    rc, _, err = run(cli, "maf2paf", few, "-o", small)
    assert rc == 0, err
    img = open(small, "rb").read()
    tab, _ = pc.bgzf_table(img)
    assert len(tab) == 2 and tab["out_len"][1] == 0 and gzip.decompress(img).count(b"\n") == 3


def test_every_command_writes_the_same_bytes_into_a_gz(cli, tmp_path):
    """utils.rs:181-228: the writer is chosen by the output's extension, whatever the command.  For each command of the path: the
    `.gz` file is a closed BGZF stream (every member checked) that inflates to exactly what the command writes to stdout;
    `.bz2` / `.xz` are refused with a message, an existing file without `-r` as well."""
    maf, paf = os.path.join(GOLDEN, "test.maf"), os.path.join(GOLDEN, "testdotplot.paf")
    cases = [("stat", maf), ("stat", "-f", "paf", paf), ("maf2paf", maf), ("maf2chain", maf), ("paf2chain", paf),
             ("call", maf, "-s", "-l0"), ("pafcov", paf), ("dotplot", "-f", "paf", paf, "--out-format", "csv"), ("dotplot", maf, "--out-format", "csv", "-m", "overview"), ("validate", paf)]
    for k, argv in enumerate(cases):
        rc, want, err = run(cli, *argv)
        assert rc == 0 and want, (argv, err)
        gz = str(tmp_path / ("out%d.txt.gz" % k))
        rc, out, err = run(cli, *argv, "-o", gz)
        assert rc == 0 and out == b"", (argv, err)
        img = open(gz, "rb").read()
        pc.bgzf_check_stream(img, want, True, one_call=False)
        rc, _, err = run(cli, *argv, "-o", gz)
        assert rc != 0 and "already exists" in err, argv
        rc, _, err = run(cli, *argv, "-o", gz, "-r")
        assert rc == 0 and open(gz, "rb").read() == img, argv          # the same command, the same bytes
    for ext in (".bz2", ".xz"):
        rc, _, err = run(cli, "stat", maf, "-o", str(tmp_path / ("o" + ext)))
        assert rc != 0 and "not built into this engine" in err
