"""Host layer (C++ `wgatools` CLI) — the parts that need no GPU: exact text formatting and the
record parsers, through the binary's hidden test hooks."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN
from wgatools_amd import build


@pytest.fixture(scope="module")
def cli():
    return build.build_cli()


def run(cli, *args, stdin=None):
    r = subprocess.run([cli] + list(args), input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout.decode(), r.stderr.decode()


def ryu_pretty_f32(x):
    """ryu 1.0.14 pretty::format32 from numpy's shortest round-trip digits (independent of the C++)"""
    x = np.float32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "-inf" if x < 0 else "inf"
    if x == 0:
        return "-0.0" if np.signbit(x) else "0.0"
    s = np.format_float_scientific(x, unique=True, trim="-")
    mant, exp = s.split("e")
    sign = "-" if mant.startswith("-") else ""
    digits = mant.lstrip("-").replace(".", "")
    exp10, n = int(exp), len(digits)
    k = exp10 - (n - 1)
    kk = n + k
    if 0 <= k and kk <= 13:
        return sign + digits + "0" * k + ".0"
    if 0 < kk <= 13:
        return sign + digits[:kk] + "." + digits[kk:]
    if -6 < kk <= 0:
        return sign + "0." + "0" * (-kk) + digits
    if n == 1:
        return sign + digits + "e" + str(kk - 1)
    return sign + digits[0] + "." + digits[1:] + "e" + str(kk - 1)


def test_f32_formatting_matches_ryu_rules(cli):
    rng = np.random.default_rng(1)
    vals = [0.99, 0.999, 1.0, 50.0, 0.84, 1e-7, 1.5e20, 123456.79, 1e13, 1e12, 9.999999e12, 1e-6,
            0.0, float("nan"), float("inf"), 3.4028235e38, 1.17549435e-38, 0.1, 16777216.0, 0.5]
    vals += rng.random(200).astype(np.float32).tolist()
    bits = [struct.unpack("<I", struct.pack("<f", np.float32(v)))[0] for v in vals]
    bits += rng.integers(0, 1 << 32, 600, dtype=np.uint64).tolist()
    rc, out, _ = run(cli, "__fmt_f32", *["%08x" % b for b in bits])
    assert rc == 0
    got = out.split("\n")[:-1]
    for b, g in zip(bits, got):
        x = np.frombuffer(struct.pack("<I", b), dtype=np.float32)[0]
        assert g == ryu_pretty_f32(x), (hex(b), g, ryu_pretty_f32(x))
    assert got[:5] == ["0.99", "0.999", "1.0", "50.0", "0.84"]
    assert got[5] == "1e-7" and got[6] == "1.5e20"


def test_natord(cli):
    pairs = [("chr2", "chr10", -1), ("chr10", "chr2", 1), ("chr1", "chr1", 0), ("x9", "x10", -1),
             ("a01", "a1", -1), ("a1", "a01", 1), ("a1b", "a1c", -1), ("ref.chr8", "ref.chr8", 0),
             ("g02#1#chr1", "g10#1#chr1", -1), ("abc", "abd", -1), ("a", "ab", -1), ("B", "a", -1)]
    args = [s for p in pairs for s in p[:2]]
    rc, out, _ = run(cli, "__natord", *args)
    assert [int(x) for x in out.split()] == [p[2] for p in pairs]


def test_cs_to_cigar(cli):
    rc, out, _ = run(cli, "__cs2cg", ":6-ata:10+gtc:4*at*tg:3", ":5=ACGT*ag:2")
    assert out.split() == ["6M3D10M3I4M2X3M", "5M1X2M"]


def test_paf_parser_semantics(cli, tmp_path):
    """csv-crate reader: '#' comment lines and blank lines skipped, CRLF ok, tags = rest"""
    p = tmp_path / "x.paf"
    p.write_bytes(b"# comment\nq1\t100\t0\t50\t+\tt1\t200\t10\t60\t40\t50\t60\tNM:i:3\tcg:Z:50M\r\n\n"
                  b"q2\t7\t1\t2\t-\tt2\t9\t3\t4\t5\t6\t0\n")
    rc, out, err = run(cli, "__parse_paf", str(p))
    assert rc == 0, err
    assert out.splitlines() == ["q1|100|0|50|+|t1|200|10|60|40|50|60|NM:i:3|cg:Z:50M",
                                "q2|7|1|2|-|t2|9|3|4|5|6|0"]
    p.write_bytes(b"q1\t100\t0\tx\t+\tt1\t200\t10\t60\t40\t50\t60\n")
    rc, out, err = run(cli, "__parse_paf", str(p))
    assert rc == 1 and "ERROR CSV deserialize error by:" in err


def test_maf_parser_semantics(cli, tmp_path):
    """first line eaten as header; a block = maximal run of lines starting with 's'"""
    rc, out, _ = run(cli, "__parse_maf", os.path.join(GOLDEN, "test.maf"))
    assert out.strip() == "block 2 [ref.chr8 181469925 1000 + 182411202 1008] [query.chr8 181989421 1007 + 183119688 1008]"
    p = tmp_path / "y.maf"
    p.write_bytes(b"s hdr 0 1 + 1 A\na score=1\ns a 0 2 + 9 AC\ns b 1 2 - 9 A-\ns c 0 1 + 5 AG\n\n# x\ns a 5 1 + 9 T\n")
    rc, out, _ = run(cli, "__parse_maf", str(p))
    assert out.splitlines() == ["block 3 [a 0 2 + 9 2] [b 1 2 - 9 2] [c 0 1 + 5 2]", "block 1 [a 5 1 + 9 1]"]
    p.write_bytes(b"#h\ns a 0 2 + 9\n")
    rc, out, err = run(cli, "__parse_maf", str(p))
    assert rc == 1 and "Parse MAF error by: S-line Filed `seq` Missing" in err


def test_unknown_subcommand_and_overwrite_guard(cli, tmp_path):
    rc, _, err = run(cli, "tview", "x")
    assert rc == 1 and "not on the CIGAR hot path" in err
    f = tmp_path / "exists.tsv"
    f.write_text("x")
    rc, _, err = run(cli, "stat", "-f", "paf", os.path.join(GOLDEN, "testdotplot.paf"), "-o", str(f))
    assert rc == 1 and "already exists, please add `-r` to rewrite it." in err


def test_line_chunk_reader_large_file_recycled_buffers(tmp_path):
    """LineChunkReader on a 90 MB file with 40 MB pieces: the multi-threaded pread path, the parallel line count and quote scan,
    buffers handed back through recycle() — every byte once and in order, pieces end at line ends"""
    import subprocess, numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "reader_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(root, "tests", "reader_check.cpp"),
                    os.path.join(root, "wgatools_amd", "host", "wga_host.cpp"), "-o", exe, "-lz", "-lpthread"], check=True)
    rng = np.random.default_rng(7)
    body = rng.integers(32, 127, 90_000_000, dtype=np.uint8)
    body[body == ord('"')] = ord("x")                     # a quote would make the reader take the rest of the file in one piece
    body[rng.integers(0, len(body), 400_000)] = 10        # lines of ~225 bytes on average, some empty
    body[-1] = 10
    path = str(tmp_path / "big.txt")
    body.tofile(path)
    want = int((np.arange(1, len(body) + 1, dtype=np.uint64) * body.astype(np.uint64)).sum(dtype=np.uint64))   # wraps mod 2^64
    out = subprocess.run([exe, path, str(40 << 20)], check=True, stdout=subprocess.PIPE).stdout.split()
    pieces, nbytes, lines, h, ends_ok = (int(x) for x in out)
    assert pieces == 3 and nbytes == len(body) and lines == int((body == 10).sum()) and ends_ok == 1
    assert h == want
    one = subprocess.run([exe, path, str(1 << 40)], check=True, stdout=subprocess.PIPE).stdout.split()     # the file in one piece
    assert int(one[0]) == 1 and int(one[3]) == want and int(one[2]) == lines
    small = subprocess.run([exe, path, str(1 << 20)], check=True, stdout=subprocess.PIPE).stdout.split()  # read(2) path, 8 MB reads
    assert int(small[3]) == want and int(small[2]) == lines and int(small[1]) == nbytes
