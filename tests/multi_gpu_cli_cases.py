"""`wgatools --gpus N` (C++ host layer, one worker thread + context per device): its files must be the bytes of the
one-device command line and of the oracle's expectation.  test_emu_cli_multi.py runs the cases on CPU (emulator build,
WGA_EMU_DEVICES devices), test_gpu_cli.py with the devices the GPU box has."""
import os
import subprocess

import numpy as np

import dist_cli_cases as dc
from parity_cases import rec_text as pc_rec_text
from wgatools_amd import synth


def run(cli, *args, env=None):
    r = subprocess.run([cli] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    return r.returncode, r.stdout, r.stderr.decode()


def msg(err):
    return err.strip().split(" ERROR ", 1)[1] if " ERROR " in err else err


def check_paf2maf(cli, tmp_path, gpus, env):
    """records over 5 target names: ordered output through per-record sizes, several pieces, several batches"""
    b = synth.make_paf_batch(91, 37, 250, 60_000)
    mapq = np.random.default_rng(2).integers(0, 61, 37)
    t_fa, q_fa, paf = dc.write_case(tmp_path, b, mapq)
    want = dc.expected_maf(b, mapq, t_fa, q_fa, 37)
    runs = [(1, None)] + [(gpus[0], c) for c in (None, "3000", "1")] + [(g, "3000") for g in gpus[1:]]
    for k, (g, chunk) in enumerate(runs):
        e = dict(env)
        if chunk:
            e["WGA_CHUNK_BYTES"] = chunk
        outp = str(tmp_path / ("o%d.maf" % k))
        rc, _, err = run(cli, "--gpus", str(g), "paf2maf", paf, "-g", t_fa, "-q", q_fa, "-o", outp, env=e)
        assert rc == 0, (g, chunk, err)
        assert open(outp, "rb").read() == want, (g, chunk)
    # not a plain file (stdout): the one-device path, same bytes
    rc, out, err = run(cli, "--gpus", str(gpus[-1]), "paf2maf", paf, "-g", t_fa, "-q", q_fa, env=env)
    assert rc == 0 and out == want, err


def check_paf2maf_errors(cli, tmp_path, gpus, env):
    """the first failing record in INPUT order ends the run on every device: the file holds the records in front of it and
    the message is that record's — an invalid base (found by the row kernel), a bad op, a tokeniser error"""
    b = synth.make_paf_batch(92, 24, 200, 40_000)
    b["strand_neg"][:] = 1
    mapq = np.zeros(24, dtype=int)
    k = 13
    pos = int(b["q_src_off"][k] + b["q_src_len"][k] // 2)
    first = min(i for i in range(24) if int(b["q_src_off"][i]) <= pos < int(b["q_src_off"][i] + b["q_src_len"][i]))
    t_fa, q_fa, paf = dc.write_case(tmp_path, b, mapq, bad_base_at=pos)
    want = dc.expected_maf(b, mapq, t_fa, q_fa, first)
    lines = open(paf).read().split("\n")
    def with_cigar(i, text, tag):
        out = list(lines)
        f = out[1 + i].split("\t")
        f[-1] = "cg:Z:" + text
        out[1 + i] = "\t".join(f)
        path = str(tmp_path / ("bad_%s.paf" % tag))
        open(path, "w").write("\n".join(out))
        return path
    cases = [(paf, first, "Invalid Base: `R`")]
    if first > 4:
        cases.append((with_cigar(4, "5=3N2=", "op"), 4, "CIGAR OP `N` invalid"))
        cases.append((with_cigar(2, "5=99999999999999999999999M", "len"), 2, "Parse `99999999999999999999999` Into Integer Error"))
    for n, (path, upto, text) in enumerate(cases):
        want_n = dc.expected_maf(b, mapq, t_fa, q_fa, upto)
        for g in sorted(set([1] + list(gpus))):
            for chunk in (None, "4000"):
                e = dict(env)
                if chunk:
                    e["WGA_CHUNK_BYTES"] = chunk
                outp = str(tmp_path / ("e%d_%d_%s.maf" % (n, g, chunk)))
                rc, _, err = run(cli, "--gpus", str(g), "paf2maf", path, "-g", t_fa, "-q", q_fa, "-o", outp, env=e)
                assert rc == 1 and msg(err) == text, (g, chunk, err)
                assert open(outp, "rb").read() == want_n, (n, g, chunk)


def check_stat(cli, tmp_path, gpus, env):
    b = synth.make_paf_batch(94, 29, 180, 30_000)
    t_fa, q_fa, paf = dc.write_case(tmp_path, b, np.zeros(29, dtype=int))
    for args in (["stat", "-f", "paf", paf], ["stat", "-f", "paf", "-e", paf]):
        ref = run(cli, *args, env=env)
        assert ref[0] == 0, ref[2]
        for g in gpus:
            for chunk in (None, "2500"):
                e = dict(env)
                if chunk:
                    e["WGA_CHUNK_BYTES"] = chunk
                got = run(cli, "--gpus", str(g), *args, env=e)
                assert got[:2] == ref[:2], (args, g, chunk, got[2])
    # a bad op in one record, a tokeniser error in an earlier one on another device: the earlier record speaks
    lines = open(paf).read().split("\n")
    def bad(pairs, tag):
        out = list(lines)
        for i, text in pairs:
            f = out[1 + i].split("\t")
            f[-1] = "cg:Z:" + text
            out[1 + i] = "\t".join(f)
        path = str(tmp_path / ("sbad_%s.paf" % tag))
        open(path, "w").write("\n".join(out))
        return path
    p1 = bad([(20, "5=3N2="), (7, "5=1Z")], "a")      # records 7 (target t2) and 20 (target t0)
    p2 = bad([(3, "5=3N2="), (11, "9M9")], "b")
    for path in (p1, p2):
        ref = run(cli, "stat", "-f", "paf", path, env=env)
        assert ref[0] == 1 and ref[1] == b""
        for g in gpus:
            got = run(cli, "--gpus", str(g), "stat", "-f", "paf", path, env=env)
            assert got[0] == 1 and got[1] == b"" and msg(got[2]) == msg(ref[2]), (path, g, got[2], ref[2])


def check_pafcov(cli, tmp_path, gpus, env):
    b = synth.make_paf_batch(93, 31, 150, 30_000)
    t_fa, q_fa, paf = dc.write_case(tmp_path, b, np.zeros(31, dtype=int))
    want = dc.expected_bed(b)
    for k, (g, chunk) in enumerate([(1, None)] + [(g, c) for g in gpus for c in (None, "2000")]):
        e = dict(env)
        if chunk:
            e["WGA_CHUNK_BYTES"] = chunk
        outp = str(tmp_path / ("c%d.bed" % k))
        rc, _, err = run(cli, "--gpus", str(g), "pafcov", paf, "-o", outp, env=e)
        assert rc == 0, (g, chunk, err)
        assert open(outp, "rb").read() == want, (g, chunk)
    # --spread: records dealt out round robin, every device holds partial counts of every target, one reduce-scatter over
    # the devices (wga_reduce_scatter_i32), every device formats the slice of the counter space it ends up with
    for g in gpus:
        for chunk in (None, "2000"):
            e = dict(env)
            if chunk:
                e["WGA_CHUNK_BYTES"] = chunk
            outp = str(tmp_path / ("cs%d%s.bed" % (g, chunk)))
            rc, _, err = run(cli, "--gpus", str(g), "--spread", "pafcov", paf, "-o", outp, env=e)
            assert rc == 0, (g, chunk, err)
            assert open(outp, "rb").read() == want, (g, chunk, "spread")
    # stdin: the input is taken whole
    outp = str(tmp_path / "cs.bed")
    r = subprocess.run([cli, "--gpus", str(gpus[-1]), "pafcov", "-o", outp], stdin=open(paf, "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0 and open(outp, "rb").read() == want, r.stderr[-500:]
    # a tokeniser error: nothing is written (buffered driver)
    lines = open(paf).read().split("\n")
    f = lines[1 + 9].split("\t")
    f[-1] = "cg:Z:5=M"
    lines[1 + 9] = "\t".join(f)
    badp = str(tmp_path / "cbad.paf")
    open(badp, "w").write("\n".join(lines))
    ref = run(cli, "pafcov", badp, env=env)
    for g in gpus:
        outp = str(tmp_path / ("cb%d.bed" % g))
        rc, _, err = run(cli, "--gpus", str(g), "pafcov", badp, "-o", outp, env=env)
        assert rc == 1 and msg(err) == msg(ref[2]), (g, err, ref[2])
        assert os.path.getsize(outp) == 0


def check_call_paf(cli, tmp_path, gpus, env):
    """`call -f paf`: every device walks and formats its records (target name hash), the rows meet in input order; a base
    outside ACGTN under a VCF row ends the run at the first such record in input order, nothing written"""
    import cli_cases as cc
    import oracle_py as orc
    b = synth.make_paf_batch(95, 31, 120, 30_000)
    t_fa, q_fa, paf = dc.write_case(tmp_path, b, np.zeros(31, dtype=int))
    tp, qp = b["t_pool"].tobytes(), b["q_pool"].tobytes()
    body = []
    for i in range(31):
        qs, ql = int(b["q_src_off"][i]), int(b["q_src_len"][i])
        ts, tl = int(b["t_src_off"][i]), int(b["t_src_len"][i])
        body.append(orc.call_within_var_paf("t%d" % (i % dc.N_T), "q%d" % (i % dc.N_Q), pc_rec_text(b, i), tp[ts:ts + tl + 1],
                                            qp[qs:qs + ql + 1], ts, ts + tl, qs, qs + ql, bool(b["strand_neg"][i]), True, 2))
    want = (cc.VCF_HEADER % "S1" + "".join(body)).encode()
    for g in (1,) + tuple(gpus):
        outp = str(tmp_path / ("c%d.vcf" % g))
        rc, _, err = run(cli, "--gpus", str(g), "call", "-f", "paf", paf, "--target", t_fa, "-q", q_fa, "-l", "2", "-s", "-n", "S1",
                         "-o", outp, "-r", env=env)
        assert rc == 0, (g, err)
        assert open(outp, "rb").read() == want, g
    # two records with a bad REF base: the earlier one in input order decides, whichever device owns it
    lines = open(paf).read().split("\n")
    xs = {}
    for i in (9, 20):
        ops = b["ops"][int(b["op_off"][i]):int(b["op_off"][i + 1])]
        k = int(np.flatnonzero((ops & 15) == 8)[0])                  # first X op: REF = the target base under it
        tb = int(sum(int(w >> 4) for w in ops[:k] if (w & 15) in (0, 7, 8, 2)))
        xs[i] = int(b["t_src_off"][i]) + tb
    tbad = bytearray(b["t_pool"].tobytes())
    tbad[xs[9]] = ord("R")
    tbad[xs[20]] = ord("Y")
    hit = sorted(i for i in range(31) for x, c in ((xs[9], "R"), (xs[20], "Y"))
                 if int(b["t_src_off"][i]) <= x <= int(b["t_src_off"][i]) + int(b["t_src_len"][i]))
    with open(t_fa, "wb") as f:
        for k in range(dc.N_T):
            f.write(b">t%d description\n" % k)
            for i in range(0, len(tbad), 60):
                f.write(bytes(tbad[i:i + 60]) + b"\n")
    os.remove(t_fa + ".fai") if os.path.exists(t_fa + ".fai") else None
    msgs = set()
    for g in (1,) + tuple(gpus):
        outp = str(tmp_path / ("e%d.vcf" % g))
        rc, _, err = run(cli, "--gpus", str(g), "call", "-f", "paf", paf, "--target", t_fa, "-q", q_fa, "-l", "2", "-s", "-o", outp,
                         "-r", env=env)
        assert rc == 1 and "invalid reference/alternate base" in err, (g, err)
        assert not os.path.exists(outp) or os.path.getsize(outp) == 0
        msgs.add(msg(err))
    assert len(msgs) == 1, msgs          # the same record's message with every device count


def check_too_many(cli, env, have):
    rc, _, err = run(cli, "--gpus", str(have + 1), "stat", "-f", "paf", "/dev/null", env=env)
    assert rc == 1 and "only %d device(s) visible" % have in err, err
