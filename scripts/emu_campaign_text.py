"""Tokeniser, the PAF / MAF splitters, the FASTA pool and the BGZF pair (K18 -> K17) on random texts the suite does not use:
python scripts/emu_campaign_text.py <first seed> <seconds>.  CPU only; results: profiles/r05_emu_campaign.txt."""
import re, sys, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import parity_cases as pc
from wgatools_amd import build, engine, _lib
eng = engine.Engine(0, _lib.load(build.EMU_LIB))
t0 = time.time(); fails = 0
def attempt(name, fn):
    global fails
    try:
        fn()
    except Exception:
        fails += 1
        print("FAIL", name); traceback.print_exc(); sys.stdout.flush()
ATOMS = [b"0", b"1", b"7", b"12", b"255", b"4096", b"268435455", b"268435456", b"4294967296", b"18446744073709551615", b"18446744073709551616",
         b"M", b"I", b"D", b"N", b"S", b"H", b"P", b"=", b"X", b"B", b"m", b"*", b" ", b"\t", b"-", b"+", b"\xc3\xa9", b"=="]
PAF_ATOMS = [b"\t", b"\n", b"\r\n", b"q", b"chr1", b"12", b"0", b"+", b"-", b"+7", b"x", b'"', b"#", b"cg:Z:5=2X", b"cs:Z::5", b"tp:A:P",
             b"18446744073709551616", b" ", b"\xc3\xa9", b"\r", b"q\t100\t0\t10\t+\tt\t200\t5\t15\t10\t10\t60\tcg:Z:10=\n"]
MAF_ATOMS = [b"\n", b"\r\n", b"s", b" ", b"\t", b"a score=1", b"ref.chr", b"12", b"+", b"-", b"ACGT-", b"x", b"\xc3\xa9", b"#", b"i", b"\x0b",
             b"a score=0\ns t 1 5 + 100 ACGT-\ns q 2 4 - 90 AC-TG\n\n"]
seed0 = int(sys.argv[1]); budget = float(sys.argv[2]); k = 0
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    rng = np.random.default_rng(s)
    texts = []
    for _ in range(int(rng.integers(1, 30))):
        t = b"".join(ATOMS[int(j)] for j in rng.integers(0, len(ATOMS), int(rng.integers(0, 25))))
        if all(int(r) < (1 << 33) or int(r) > 0xFFFFFFFFFFFFFFFF for r in re.findall(rb"[0-9]+", t)):
            texts.append(t)
    if texts:
        attempt("tokeniser %d" % s, lambda: pc.check_tokeniser(eng, texts))
    paf = b"".join(PAF_ATOMS[int(j)] for j in rng.integers(0, len(PAF_ATOMS), int(rng.integers(0, 60))))
    attempt("paf_split %d" % s, lambda: pc.check_paf_split(eng, paf))
    maf = b"".join(MAF_ATOMS[int(j)] for j in rng.integers(0, len(MAF_ATOMS), int(rng.integers(0, 60))))
    attempt("maf_split %d" % s, lambda: pc.check_maf_split(eng, maf))
    attempt("fasta %d" % s, lambda: pc.check_fasta_pool(eng, pc.random_fasta(rng, int(rng.integers(1, 12)), int(rng.integers(1, 30000)), crlf=bool(s & 1))))
    # K18 -> K17 on text of a random alphabet
    n = int(rng.choice([0, 1, 127, 129, 32767, 32768, 32769, 70000])) + int(rng.integers(0, 3))
    kk = int(rng.integers(1, 257))
    p = rng.random(kk) ** int(rng.integers(1, 8)) + 1e-9
    data = rng.permutation(256)[:kk][rng.choice(kk, n, p=p / p.sum())].astype(np.uint8).tobytes()
    def bg():
        d_in = eng.upload(np.frombuffer(data + b"\0" * 8, dtype=np.uint8))
        out, used = eng.bgzf_compress(d_in, n, eof_marker=bool(s & 2))
        img = out.numpy()[:used].tobytes()
        pc.bgzf_check_stream(img, data, bool(s & 2))
        if n:
            tab, total = pc.bgzf_table(img)
            back = eng.empty(total + 16, np.uint8).fill(0x23)
            st = eng.empty(len(tab), np.uint32).fill(0xFF)
            eng.bgzf_inflate(eng.upload(np.frombuffer(img + b"\0" * 16, dtype=np.uint8)), len(img), len(tab), eng.upload(tab.view(np.uint8)), back, st)
            assert (st.numpy() == 0).all() and back.numpy()[:total].tobytes() == data
    attempt("bgzf %d" % s, bg)
    if k % 20 == 0:
        print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
print("END", k, "seeds", fails, "fails")
