"""Cases for the multi-rank harness (tests/dist_cli.py over wgatools_amd/multigpu.py): its files must be the bytes the single-GPU `wgatools`
command line writes, whatever the number of ranks.  test_dist_cli_gloo.py runs them on CPU (emulator build of the kernels,
1 and 2 ranks over gloo), test_gpu_cli.py on the GPU box (one rank, libwgahip.so)."""
import json
import os
import subprocess
import sys

import numpy as np

import oracle_py as orc
import parity_cases as pc
from wgatools_amd import engine, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_T, N_Q = 5, 3


def write_case(tmp_path, b, mapq, bad_base_at=None):
    """PAF + FASTA files: record i aligns query contig q<i % 3> to target contig t<i % 5> (every contig = the pool)"""
    t_fa, q_fa, paf = tmp_path / "t.fa", tmp_path / "q.fa", tmp_path / "in.paf"
    qp = b["q_pool"].copy()
    if bad_base_at is not None:
        qp[bad_base_at] = ord("R")
    def fasta(path, names, seq):
        with open(path, "wb") as f:
            for nm in names:
                f.write(b">" + nm.encode() + b" description\n")
                for i in range(0, len(seq), 60):
                    f.write(seq[i:i + 60] + b"\n")
    fasta(t_fa, ["t%d" % k for k in range(N_T)], b["t_pool"].tobytes())
    fasta(q_fa, ["q%d" % k for k in range(N_Q)], qp.tobytes())
    n = len(b["strand_neg"])
    with open(paf, "w") as f:
        f.write("# synthetic\n")
        for i in range(n):
            qs, ql = int(b["q_src_off"][i]), int(b["q_src_len"][i])
            ts, tl = int(b["t_src_off"][i]), int(b["t_src_len"][i])
            f.write("q%d\t%d\t%d\t%d\t%s\tt%d\t%d\t%d\t%d\t%d\t%d\t%d\tNM:i:0\t%s\n" % (
                i % N_Q, len(b["q_pool"]), qs, qs + ql, "-" if b["strand_neg"][i] else "+", i % N_T, len(b["t_pool"]), ts,
                ts + tl, 0, 0, mapq[i], pc.rec_text(b, i)))
    return str(t_fa), str(q_fa), str(paf)


def expected_maf(b, mapq, t_fa, q_fa, upto):
    out = ["#maf version=1.6 convert_from=paf t_seq_path=%s q_seq_path=%s\n" % (t_fa, q_fa)]
    for i in range(upto):
        et, eq = pc.oracle_rows(b, i)
        qs, ql = int(b["q_src_off"][i]), int(b["q_src_len"][i])
        ts, tl = int(b["t_src_off"][i]), int(b["t_src_len"][i])
        neg = bool(b["strand_neg"][i])
        out.append("a score=%d\ns\tt%d\t%d\t%d\t+\t%d\t%s\ns\tq%d\t%d\t%d\t%s\t%d\t%s\n\n" % (
            mapq[i], i % N_T, ts, tl, len(b["t_pool"]), et.decode(), i % N_Q, len(b["q_pool"]) - (qs + ql) if neg else qs, ql,
            "-" if neg else "+", len(b["q_pool"]), eq.decode()))
    return "".join(out).encode()


def expected_bed(b):
    """pafcov.rs:18-64 through the oracle: targets in first-appearance order, one line per position"""
    n = len(b["strand_neg"])
    L = len(b["t_pool"])
    cov = {}
    order = []
    for i in range(n):
        t = "t%d" % (i % N_T)
        if t not in cov:
            cov[t] = np.zeros(L, dtype=np.uint64)
            order.append(t)
        orc.update_cov_vec(cov[t], pc.rec_text(b, i), int(b["t_src_off"][i]))
    return "".join("%s\t%d\t%d\t%d\n" % (t, p, p + 1, int(cov[t][p])) for t in order for p in range(L)).encode()


def launch(world, lib, port, *args, expect_rc=0):
    cmd = [sys.executable]
    if world > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "tests", "dist_cli.py")] + (["--lib", lib] if lib else []) + list(args)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == expect_rc or (world > 1 and expect_rc and r.returncode != 0), (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    return r


def check_paf2maf(tmp_path, lib, cli, worlds, port):
    b = synth.make_paf_batch(91, 37, 250, 60_000)
    mapq = np.random.default_rng(2).integers(0, 61, 37)
    t_fa, q_fa, paf = write_case(tmp_path, b, mapq)
    want = expected_maf(b, mapq, t_fa, q_fa, 37)
    ref = str(tmp_path / "ref.maf")
    r = subprocess.run([cli, "paf2maf", paf, "-g", t_fa, "-q", q_fa, "-o", ref], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and open(ref, "rb").read() == want, r.stderr[-500:]
    for w in worlds:
        outp = str(tmp_path / ("out%d.maf" % w))
        launch(w, lib, port + w, "paf2maf", paf, "-g", t_fa, "-q", q_fa, "-o", outp, "--chunk-bytes", "20000")   # several resident batches per rank
        assert open(outp, "rb").read() == want, w


def check_paf2maf_error(tmp_path, lib, worlds, port):
    """an invalid base in the query slice of a '-' strand record: records in front of it are written, every rank stops"""
    b = synth.make_paf_batch(92, 24, 200, 40_000)
    b["strand_neg"][:] = 1
    mapq = np.zeros(24, dtype=int)
    k = 13
    pos = int(b["q_src_off"][k] + b["q_src_len"][k] // 2)
    later = [i for i in range(24) if int(b["q_src_off"][i]) <= pos < int(b["q_src_off"][i] + b["q_src_len"][i])]
    first = min(later)                                     # slices of other records may hold the position as well
    t_fa, q_fa, paf = write_case(tmp_path, b, mapq, bad_base_at=pos)
    want = expected_maf(b, mapq, t_fa, q_fa, first)
    for w in worlds:
        outp = str(tmp_path / ("err%d.maf" % w))
        r = launch(w, lib, port + w, "paf2maf", paf, "-g", t_fa, "-q", q_fa, "-o", outp, "--chunk-bytes", "9000", expect_rc=1)
        assert "Invalid Base: `R`" in r.stderr, r.stderr[-1500:]
        got = open(outp, "rb").read()
        assert got[:len(want)] == want and not got[len(want):].strip(b"\x00"), (w, first, len(got), len(want))


def check_pafcov(tmp_path, lib, cli, worlds, port):
    b = synth.make_paf_batch(93, 31, 150, 30_000)
    t_fa, q_fa, paf = write_case(tmp_path, b, np.zeros(31, dtype=int))
    want = expected_bed(b)
    r = subprocess.run([cli, "pafcov", paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == want, r.stderr[-500:]
    for w in worlds:
        for spread in (False, True):
            outp = str(tmp_path / ("cov%d%d.bed" % (w, spread)))
            launch(w, lib, port + 10 + 2 * w + spread, "pafcov", paf, "-o", outp, "--chunk-bytes", "15000", *(["--spread"] if spread else []))
            assert open(outp, "rb").read() == want, (w, spread)


def check_totals(tmp_path, lib, worlds, port):
    b = synth.make_paf_batch(94, 29, 180, 30_000)
    t_fa, q_fa, paf = write_case(tmp_path, b, np.zeros(29, dtype=int))
    exp = np.zeros(11, dtype=np.int64)
    for i in range(29):
        exp += np.array(orc.parse_paf_to_cigar(pc.rec_text(b, i), b["strand_neg"][i]), dtype=np.int64)
    for w in worlds:
        r = launch(w, lib, port + 20 + w, "totals", paf, "--chunk-bytes", "12000")
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        assert [d[k] for k in engine.COUNTS_DTYPE.names] == exp.tolist() and d["records"] == 29 and d["ranks"] == w, d


def check_bad_cigar_ends_every_rank(tmp_path, lib, worlds, port):
    """a CIGAR the tokeniser rejects (a length beyond u64) in one record: only the rank that owns the record sees it, yet
    `pafcov` (sharded and spread) and `totals` end with status 1 on every rank — no collective is left waiting — and the
    owner prints the reference's message; an op `stat` rejects ends `totals` the same way"""
    b = synth.make_paf_batch(95, 23, 120, 30_000)
    t_fa, q_fa, paf = write_case(tmp_path, b, np.zeros(23, dtype=int))
    lines = open(paf).read().split("\n")
    def with_cigar(k, text):
        out = list(lines)
        f = out[1 + k].split("\t")
        f[-1] = "cg:Z:" + text
        out[1 + k] = "\t".join(f)
        path = str(tmp_path / ("bad_%d_%d.paf" % (k, len(text))))
        open(path, "w").write("\n".join(out))
        return path
    bad_len = with_cigar(11, "5=99999999999999999999999M3=")
    bad_op = with_cigar(7, "5=3N2=")
    for w in worlds:
        for k, extra in enumerate(([], ["--spread"])):
            r = launch(w, lib, port + 4 * w + k, "pafcov", bad_len, "-o", str(tmp_path / ("bad%d%d.bed" % (w, k))), *extra, expect_rc=1)
            assert "Parse `99999999999999999999999` Into Integer Error" in r.stderr, r.stderr[-1500:]
        r = launch(w, lib, port + 4 * w + 2, "totals", bad_len, expect_rc=1)
        assert "Parse `99999999999999999999999` Into Integer Error" in r.stderr, r.stderr[-1500:]
        r = launch(w, lib, port + 4 * w + 3, "totals", bad_op, expect_rc=1)
        assert "CIGAR OP `N` invalid" in r.stderr, r.stderr[-1500:]
