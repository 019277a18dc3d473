"""The C oracle AND the HIP path against a second, independent restatement of cigar.rs.

tests/golden/py_restatement.json is written by scripts/make_golden_py.py: Python versions of the tokeniser fold,
parse_paf_to_cigar, update_cov_vec, gen_pesudo_maf_by_cigar, parse_cigar_to_insert and reverse_complement, written from the
Rust text alone (not from oracle/oracle.c), run on the reference's demo PAF and on 200 seeded adversarial CIGARs.  Nothing
in the reference pins those functions, so this is the next best thing: two readers of the same Rust text, and
the kernels, have to agree on every case — results, error kinds, the quoted token, and where the reference would panic.

* `-m "not gpu"`: the oracle against the file, and the kernel source on the SIMT emulator against the file;
* `-m gpu`: libwgahip.so against the file.
The engine checks reuse the parity helpers (tests/parity_cases.py) with their oracle replaced by a look-up into the file, so
the oracle takes no part in them."""
import json
import os
import types

import numpy as np
import pytest

import oracle_py as orc
import parity_cases as pc
from helpers import GOLDEN, pack_records

KIND = {"CigarOpInvalid": 2, "ParseIntError": 3, "InvalidBase": 4, "Nom": 5}


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLDEN, "py_restatement.json")) as f:
        d = json.load(f)
    assert len(d["cases"]) >= 200
    return d["cases"]


def expect(outcome, fn, what):
    """fn() must give outcome["ok"], or raise the OracleError the outcome names (kind, quoted argument; a panic is kind 6)"""
    if "ok" in outcome:
        got = fn()
        assert got == outcome["ok"], (what, got if len(str(got)) < 200 else str(got)[:200], str(outcome["ok"])[:200])
        return
    with pytest.raises(orc.OracleError) as ei:
        fn()
    e = ei.value
    if "panic" in outcome:
        assert e.kind == 6, (what, e.kind, e.message)
    else:
        assert e.kind == KIND[outcome["err"]], (what, e.kind, e.message, outcome)
        assert e.arg.encode() == outcome["arg"].encode()[:63], (what, e.arg, outcome["arg"])


# ---------------------------------------------------------------------------------------------------------------------------
# the C oracle against the file
# ---------------------------------------------------------------------------------------------------------------------------
def test_oracle_tokens(gold):
    n = 0
    for c in gold:
        cg = c["cg"]
        if not cg.startswith("cg:Z:") or cg == "cg:Z:":
            continue                                      # orc.tokenise takes the text behind the tag; tags: test_oracle_stat
        toks, kind, (eo, el) = orc.tokenise(cg[5:])
        o = c["tokens"]
        if "ok" in o:
            assert kind == 0, (cg[:60], kind)
            assert toks == [(ln, op.encode()[0]) for ln, op in o["ok"]], cg[:60]
        else:
            assert kind == KIND[o["err"]], (cg[:60], kind, o)
            assert cg[5:].encode()[eo:eo + el] == o["arg"].encode(), (cg[:60], eo, el, o)
        n += 1
    assert n >= 180


def test_oracle_stat(gold):
    for c in gold:
        o = c["stat"]
        want = {"ok": tuple(o["ok"])} if "ok" in o else o
        expect(want, lambda: orc.parse_paf_to_cigar(c["cg"], c["neg"]), c["cg"][:60])


def test_oracle_cov(gold):
    for c in gold:
        cv = c["cov"]

        def run():
            cov = np.zeros(cv["tlen"], dtype=np.uint64)
            orc.update_cov_vec(cov, c["cg"], cv["start"])
            return cov
        o = cv["runs"]
        if "ok" in o:
            exp = np.zeros(cv["tlen"], dtype=np.uint64)
            for a, b in o["ok"]:
                exp[a:b] += 1
            assert (run() == exp).all(), c["cg"][:60]
        else:
            expect(o, run, c["cg"][:60])


def test_oracle_insert_and_pseudo(gold):
    n_ins = n_ps = 0
    for c in gold:
        if "insert" in c:
            ins = c["insert"]

            def rows():
                q = ins["q_fwd"].encode()
                if c["neg"]:
                    q = orc.reverse_complement(q)
                t, q = orc.parse_cigar_to_insert(c["cg"], ins["t"].encode(), q)
                return [t.decode(), q.decode()]
            expect(ins["rows"], rows, c["cg"][:60])
            n_ins += 1
        if "pseudo_base" in c:
            pb = c["pseudo_base"]

            def row():
                q = pb["q_fwd"].encode()
                if c["neg"]:
                    q = orc.reverse_complement(q)
                return orc.gen_pesudo_maf_by_cigar(c["cg"], q, True).decode()
            expect(pb["row"], row, c["cg"][:60])
            n_ps += 1
        if c.get("pseudo_sym") is not None:
            expect(c["pseudo_sym"], lambda: orc.gen_pesudo_maf_by_cigar(c["cg"], b"", False).decode(), c["cg"][:60])
    assert n_ins >= 150 and n_ps >= 150


# ---------------------------------------------------------------------------------------------------------------------------
# the kernels against the file: the parity helpers with a look-up in the oracle's place
# ---------------------------------------------------------------------------------------------------------------------------
class GoldenLookup:
    """stands where parity_cases expects the oracle module: the functions answer from the file, keyed by their arguments"""
    OracleError = orc.OracleError

    def __init__(self):
        self.stat, self.rows, self.cov, self.pseudo, self.tok = {}, {}, {}, {}, {}
        self.asked = 0

    @staticmethod
    def _give(o):
        if "ok" in o:
            return o["ok"]
        if "panic" in o:
            raise orc.OracleError(6, "", "panic (golden)")
        raise orc.OracleError(KIND[o["err"]], o["arg"], "%s (golden)" % o["err"])

    def parse_paf_to_cigar(self, text, neg):
        self.asked += 1
        return tuple(self._give(self.stat[(text, int(neg))]))

    def reverse_complement(self, seq):        # utils.rs:83-101 as the generator restates it
        comp = dict(zip(b"ACGTNacgtn", b"TGCANtgcan"))
        out = bytearray()
        for ch in reversed(bytes(seq)):
            if ch not in comp:
                raise orc.OracleError(4, chr(ch), "InvalidBase (golden)")
            out.append(comp[ch])
        return bytes(out)

    def parse_cigar_to_insert(self, text, t, q):
        self.asked += 1
        t_row, q_row = self._give(self.rows[(text, bytes(t), bytes(q))])
        return t_row.encode(), q_row.encode()

    def update_cov_vec(self, cov, text, start):
        self.asked += 1
        for a, b in self._give(self.cov[(text, int(start), len(cov))]):
            cov[a:b] += 1

    def gen_pesudo_maf_by_cigar(self, text, q, base):
        self.asked += 1
        return self._give(self.pseudo[(text, bytes(q), bool(base))]).encode()

    def tokenise(self, text):
        self.asked += 1
        return self.tok[bytes(text)]


def packable(eng, gold, need=None):
    """the cases whose CIGAR the boundary's packer accepts (all tokens well-formed), as a batch; optional: only cases
    holding `need`"""
    sel = [c for c in gold if "ok" in c["tokens"] and (need is None or c.get(need) is not None)]
    # op chars the packed form names; a length of 2^40 would be 4 096 packed ops of 2^28 - 1 and 2^64 - 1 a terabyte of them
    sel = [c for c in sel if all(op in "MIDNSHP=X" and ln < (1 << 33) for ln, op in c["tokens"]["ok"])]
    ops, off, errs = pack_records(eng, [c["cg"][5:] for c in sel])
    assert all(e == 0 for e in errs)
    return sel, dict(ops=ops, op_off=off, strand_neg=np.array([c["neg"] for c in sel], dtype=np.uint8))


@pytest.fixture()
def lookup(monkeypatch):
    lk = GoldenLookup()
    monkeypatch.setattr(pc, "orc", lk)
    return lk


def engine_stat(eng, gold, lk):
    sel, b = packable(eng, gold)
    for i, c in enumerate(sel):
        lk.stat[(pc.rec_text(b, i), c["neg"])] = c["stat"]
    pc.check_stat(eng, b)
    assert lk.asked == len(sel) >= 130


def engine_cov(eng, gold, lk):
    sel, b = packable(eng, gold)
    sel_idx = [i for i, c in enumerate(sel) if c["cov"]["tlen"] <= 1 << 20]
    # every case is a target of its own (its length and start are the fixture's)
    keep = np.array(sel_idx)
    ops = np.concatenate([pc.rec_ops(b, i) for i in keep]).astype(np.uint32)
    off = np.cumsum([0] + [len(pc.rec_ops(b, i)) for i in keep]).astype(np.uint64)
    bb = dict(ops=ops, op_off=off, strand_neg=b["strand_neg"][keep])
    for j, i in enumerate(keep):
        cv = sel[i]["cov"]
        lk.cov[(pc.text_any(pc.rec_ops(bb, j)), cv["start"], cv["tlen"])] = cv["runs"]
    pc.check_pafcov(eng, bb, list(range(len(keep))), [sel[i]["cov"]["start"] for i in keep],
                    [sel[i]["cov"]["tlen"] for i in keep], align=1)
    assert lk.asked == len(keep) >= 120


def engine_insert(eng, gold, lk):
    sel, _ = packable(eng, gold, need="insert")
    b = pc.batch_from_texts(eng, [c["cg"][5:] for c in sel], [c["neg"] for c in sel],
                            [c["insert"]["t"].encode() for c in sel], [c["insert"]["q_fwd"].encode() for c in sel], pad=40)
    for i, c in enumerate(sel):
        q = c["insert"]["q_fwd"].encode()
        try:
            q = lk.reverse_complement(q) if c["neg"] else q
        except orc.OracleError:
            continue                      # InvalidBase: check_paf2maf stops at the reverse complement, as the driver does
        lk.rows[(pc.rec_text(b, i), c["insert"]["t"].encode(), q)] = c["insert"]["rows"]
    pc.check_paf2maf(eng, b)
    assert lk.asked >= 110


def engine_pseudo(eng, gold, lk, base_mode):
    sel, _ = packable(eng, gold, need="pseudo_base" if base_mode else "pseudo_sym")
    qs = [c["pseudo_base"]["q_fwd"].encode() if base_mode else b"" for c in sel]
    b = pc.batch_from_texts(eng, [c["cg"][5:] for c in sel], [c["neg"] for c in sel], [b"" for _ in sel], qs, pad=40)
    for i, c in enumerate(sel):
        q = qs[i]
        if base_mode and c["neg"]:
            q = lk.reverse_complement(q)                    # the fixture's pseudo queries hold ACGTN only
        lk.pseudo[(pc.text_any(pc.rec_ops(b, i)), q, bool(base_mode))] = c["pseudo_base"]["row"] if base_mode else c["pseudo_sym"]
    pc.check_pafpseudo(eng, b, base_mode)
    assert lk.asked >= 110


def engine_tokens(eng, gold, lk):
    texts = []
    for c in gold:
        cg = c["cg"]
        if not cg.startswith("cg:Z:") or cg == "cg:Z:":
            continue
        t = cg[5:].encode()
        o = c["tokens"]
        if "ok" in o:
            if any(ln >= (1 << 33) for ln, _ in o["ok"]):
                continue                        # gigabytes of packed ops (see packable)
            lk.tok[t] = ([(ln, op.encode()[0]) for ln, op in o["ok"]], 0, (0, 0))
        else:
            # the tokens in front of the failing one are whatever the packer kept: only kind and quoted token are compared
            at = t.find(o["arg"].encode()) if o["arg"] else -1
            toks, kind, span = orc.tokenise(t)     # token prefix: not part of the golden (the fold yields none on an error)
            assert kind == KIND[o["err"]]
            if o["arg"]:
                assert t[span[0]:span[0] + span[1]] == o["arg"].encode() and at >= 0
            lk.tok[t] = (toks, KIND[o["err"]], span)
        texts.append(t)
    pc.check_tokeniser(eng, texts)
    assert lk.asked == len(texts) >= 170


def test_emu_stat(emu, gold, lookup):
    engine_stat(emu, gold, lookup)


def test_emu_cov(emu, gold, lookup):
    engine_cov(emu, gold, lookup)


def test_emu_insert(emu, gold, lookup):
    engine_insert(emu, gold, lookup)


@pytest.mark.parametrize("base_mode", [1, 0])
def test_emu_pseudo(emu, gold, lookup, base_mode):
    engine_pseudo(emu, gold, lookup, base_mode)


def test_emu_tokens(emu, gold, lookup):
    engine_tokens(emu, gold, lookup)


@pytest.mark.gpu
def test_gpu_stat(gpu, gold, lookup):
    engine_stat(gpu, gold, lookup)


@pytest.mark.gpu
def test_gpu_cov(gpu, gold, lookup):
    engine_cov(gpu, gold, lookup)


@pytest.mark.gpu
def test_gpu_insert(gpu, gold, lookup):
    engine_insert(gpu, gold, lookup)


@pytest.mark.gpu
@pytest.mark.parametrize("base_mode", [1, 0])
def test_gpu_pseudo(gpu, gold, lookup, base_mode):
    engine_pseudo(gpu, gold, lookup, base_mode)


@pytest.mark.gpu
def test_gpu_tokens(gpu, gold, lookup):
    engine_tokens(gpu, gold, lookup)
