"""A SECOND restatement of the unpinned CIGAR functions, written from the Rust text alone, and the fixtures it generates.

The C oracle (oracle/oracle.c) is one reader's restatement of cigar.rs; nothing the reference holds pins its stat, pafcov,
pafpseudo and paf2maf functions (there is no Rust toolchain here).  This file holds Python versions of the same
functions written from /root/reference/src/parser/cigar.rs, src/utils.rs and src/errors.rs WITHOUT looking at oracle.c, and
writes their results on the reference's demo PAF (tests/golden/testdotplot.paf) and on 200 seeded adversarial CIGARs to
tests/golden/py_restatement.json.  tests/test_golden_py.py then checks BOTH the C oracle and the HIP path against that file:
a misreading of the Rust text has to be made twice, independently, to go unnoticed.  It pins nothing to the reference itself.

Integer arithmetic wraps modulo 2^64 as in the release build the reference ships (its Cargo.toml has no overflow-checks).

Run in the build container only (python scripts/make_golden_py.py); the JSON travels, this script is not imported by tests.

Restated (file:line in /root/reference/src):
  tokens            parser/cigar.rs:43-75 (cst2cu, parse_cigar_str_tuple) under nom::fold_many1 with the Result accumulator
                    of every caller (first error sticks, later units are skipped), utils.rs:69-74 (parse_str2u64),
                    errors.rs:88-97 (the nom error keeps input[..10]: a panic for shorter inputs)
  parse_paf_to_cigar      parser/cigar.rs:629-707
  update_cov_vec          parser/cigar.rs:710-741
  gen_pesudo_maf_by_cigar parser/cigar.rs:744-804
  parse_cigar_to_insert   parser/cigar.rs:492-551
  reverse_complement      utils.rs:83-101
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


W = (1 << 64) - 1      # usize / u64 arithmetic of a release build wraps


class WGAError(Exception):
    def __init__(self, kind, arg):
        super().__init__("%s(%r)" % (kind, arg))
        self.kind, self.arg = kind, arg


class Panic(Exception):
    pass


def nom_error(rest):
    """errors.rs:88-97: NomErr(input[..10]) — slicing a str of fewer than 10 bytes (or inside a character) panics"""
    b = rest.encode()
    if len(b) < 10:
        raise Panic("byte index 10 is out of bounds")
    try:
        head = b[:10].decode()
    except UnicodeDecodeError:
        raise Panic("byte index 10 is not a char boundary")
    raise WGAError("Nom", head)


def is_digit(c):
    return "0" <= c <= "9"


def fold_units(cigar, unit):
    """tag("cg:Z:") then fold_many1(parse_cigar_str_tuple, null, closure): `unit(op, length)` is the closure's body behind
    cst2cu; once the accumulator is an Err the remaining units are parsed and ignored"""
    if not cigar.startswith("cg:Z:"):
        nom_error(cigar)
    s = cigar[5:]
    if s == "":
        nom_error(s)            # the parser fails on the first application: fold_many1's own error, on the empty input
    err = None
    i = 0
    while i < len(s):           # the parser consumes at least one character of any non-empty input
        j = i
        while j < len(s) and is_digit(s[j]):
            j += 1
        k = j
        while k < len(s) and not is_digit(s[k]):
            k += 1
        digits, op = s[i:j], s[j:k]
        i = k
        if err is None:
            try:
                if len(op) != 1:                                   # cst2cu: no char, or a second one
                    raise WGAError("CigarOpInvalid", op)
                if digits == "" or int(digits) >= 1 << 64:          # "".parse::<u64>() fails; so does an overflow
                    raise WGAError("ParseIntError", digits)
                unit(op, int(digits))
            except WGAError as e:
                err = e
    if err is not None:
        raise err


def tokens(cigar):
    out = []
    fold_units(cigar, lambda op, n: out.append([n, op]))
    return out


def parse_paf_to_cigar(cigar, negative):
    c = dict(match=0, mismatch=0, ins_ev=0, ins_bp=0, del_ev=0, del_bp=0, inv_ins_ev=0, inv_ins_bp=0, inv_del_ev=0,
             inv_del_bp=0, inv_ev=1 if negative else 0)

    def unit(op, n):
        if op in "M=":
            c["match"] += n
        elif op == "X":
            c["mismatch"] += n
        elif op == "I":
            c["inv_ins_ev" if negative else "ins_ev"] += 1
            c["inv_ins_bp" if negative else "ins_bp"] += n
        elif op == "D":
            c["inv_del_ev" if negative else "del_ev"] += 1
            c["inv_del_bp" if negative else "del_bp"] += n
        else:
            raise WGAError("CigarOpInvalid", op)
    fold_units(cigar, unit)
    return [c[k] % (1 << 64) for k in ("match", "mismatch", "ins_ev", "ins_bp", "del_ev", "del_bp", "inv_ins_ev",
                                       "inv_ins_bp", "inv_del_ev", "inv_del_bp", "inv_ev")]


def update_cov_vec(tlen, cigar, start):
    """-> the covered runs [a, b) below tlen, in op order (cov_vec[i] += 1 for i in them)"""
    runs = []
    pos = [start]

    def unit(op, n):
        if op in "M=":
            a, b = pos[0], min((pos[0] + n) & W, tlen)      # `for i in pos..(pos + length)`: empty when the sum wrapped
            if a < b:
                runs.append([a, b])
            pos[0] = (pos[0] + n) & W
        elif op in "IS":
            pass
        else:
            pos[0] = (pos[0] + n) & W
    fold_units(cigar, unit)
    return runs


def gen_pesudo_maf_by_cigar(cigar, raw_q_seq, base):
    seq = [raw_q_seq]
    off = [0]

    def unit(op, n):
        s = seq[0]
        if op in "M=":
            if base:
                off[0] = (off[0] + n) & W
            else:
                seq[0] = s + "1" * n
        elif op in "IS":
            if base:
                end = (off[0] + n) & W
                if end > len(s) or off[0] > end:
                    raise Panic("String::drain beyond the end")
                seq[0] = s[:off[0]] + s[off[0] + n:]
        elif op == "D":
            if base:
                if off[0] > len(s):
                    raise Panic("String::insert_str beyond the end")
                seq[0] = s[:off[0]] + "-" * n + s[off[0]:]
                off[0] = (off[0] + n) & W
            else:
                seq[0] = s + "-" * n
        elif op == "X":
            if base:
                off[0] = (off[0] + n) & W
            else:
                seq[0] = s + "0" * n
    fold_units(cigar, unit)
    return seq[0]


def parse_cigar_to_insert(cigar, t_seq, q_seq):
    t, q, off = [t_seq], [q_seq], [0]

    def unit(op, n):
        if op in "M=X":
            off[0] = (off[0] + n) & W
        elif op == "I":
            if off[0] > len(t[0]):
                raise Panic("String::insert_str beyond the end")
            t[0] = t[0][:off[0]] + "-" * n + t[0][off[0]:]
            off[0] = (off[0] + n) & W
        elif op == "D":
            if off[0] > len(q[0]):
                raise Panic("String::insert_str beyond the end")
            q[0] = q[0][:off[0]] + "-" * n + q[0][off[0]:]
            off[0] = (off[0] + n) & W
        else:
            raise WGAError("CigarOpInvalid", op)
    fold_units(cigar, unit)
    return t[0], q[0]


COMP = dict(zip("ACGTNacgtn", "TGCANtgcan"))


def reverse_complement(s):
    out = []
    for c in reversed(s):
        if c not in COMP:
            raise WGAError("InvalidBase", c)
        out.append(COMP[c])
    return "".join(out)


# ------------------------------------------------------------------------------------------------------------------------
def outcome(fn):
    try:
        return {"ok": fn()}
    except WGAError as e:
        return {"err": e.kind, "arg": e.arg}
    except Panic:
        return {"panic": 1}


def consumption(cigar):
    """(target bases, query bases of paf2maf, query bases of pafpseudo) the well-formed units of a CIGAR consume"""
    t = q = qp = 0
    try:
        for n, op in tokens(cigar):
            if op in "M=X":
                t, q, qp = t + n, q + n, qp + n
            elif op == "I":
                q, qp = q + n, qp + n
            elif op == "D":
                t += n
            elif op == "S":
                qp += n
    except (WGAError, Panic):
        pass
    return t, q, qp


def rand_seq(rng, n, dirty=False):
    s = "".join(rng.choice("ACGTacgtNn") for _ in range(n))
    if dirty and n:
        k = rng.randrange(n)
        s = s[:k] + rng.choice("RYKM-*x") + s[k + 1:]
    return s


def adversarial_cigars(rng, n):
    """well-formed CIGARs of every op mix (clips, N / H / P, zero lengths, long runs of indels), then the malformed ones the
    tokeniser has rules for (multi-char and multi-byte ops, missing lengths, trailing digits, overflowing lengths, a missing or
    wrong tag, the empty CIGAR) and lengths of 2^28 and more"""
    out = []
    plain = "M=X" * 3 + "ID"
    for k in range(n):
        kind = k % 10
        units = []
        m = rng.choice([1, 2, 3, 5, 9, 17, 40])
        if kind <= 4:                                   # well-formed, M = X I D only
            for _ in range(m):
                op = rng.choice(plain)
                ln = rng.choice([0, 1, 1, 2, 3, 7, 16, 33, 100]) if op in "ID" else rng.choice([1, 2, 5, 16, 24, 60, 90])
                units.append("%d%s" % (ln, op))
            if kind == 4:                               # runs of indels back to back, also at both ends
                units = ["3I", "2D"] + units + ["4D", "1I", "0I"]
        elif kind == 5:                                 # clips and the ops stat rejects
            units = [rng.choice(["5S", "12H", "0S"])]
            for _ in range(m):
                units.append("%d%s" % (rng.choice([0, 1, 4, 30, 90]), rng.choice("M=XIDNP" if rng.random() < 0.5 else "M=XID")))
            units.append(rng.choice(["7S", "3H", "1P"]))
        elif kind == 6:                                 # leading zeros, upper / lower case, ops outside the alphabet
            for _ in range(m):
                units.append("%s%s" % (rng.choice(["007", "0", "00", "15", "1"]), rng.choice("M=XIDmxBZ*")))
        elif kind == 7:                                 # malformed tokens
            base = ["10M", "2I", "5=", "3D", "4X"]
            rng.shuffle(base)
            bad = rng.choice(["MM", "5MM", "M", "=5", "12", "3M12", "5\u00e9", "5M\u00e9\u00e9", "18446744073709551616M",
                              "99999999999999999999999I", "4 M", "5M\t"])
            pos = rng.randrange(len(base) + 1)
            units = base[:pos] + [bad] + base[pos:]
        elif kind == 8:                                 # lengths of 2^28 and beyond (no sequences go with these)
            big = rng.choice([(1 << 28) - 1, 1 << 28, (1 << 28) + 5, (1 << 32) + 7, (1 << 40), (1 << 64) - 1])
            units = ["10=", "%d%s" % (big, rng.choice("MDNI=X")), "7=", "2X", "%d%s" % (rng.choice([1, big]), rng.choice("DIS"))]
        else:                                           # tags
            body = "".join("%d%s" % (rng.choice([1, 9, 50]), rng.choice("M=XID")) for _ in range(m))
            out.append(rng.choice(["cg:Z:", "cg:Z", "", "cs:Z::10", "CG:Z:" + body, "cg:z:" + body, " cg:Z:" + body,
                                   "cg:Z:" + body + " ", "xx", "cg:Z:\u00e9\u00e9\u00e9\u00e9", "0123456789", "cg:A:" + body]))
            continue
        out.append("cg:Z:" + "".join(units))
    return out


def make_case(rng, cigar, negative, with_seqs=True):
    t_need, q_need, qp_need = consumption(cigar)
    case = {"cg": cigar, "neg": int(negative), "tokens": outcome(lambda: tokens(cigar)),
            "stat": outcome(lambda: parse_paf_to_cigar(cigar, negative))}
    tlen = max(1, t_need + rng.choice([-40, -1, 0, 0, 1, 300]))
    if t_need >= 1 << 27:
        tlen = rng.choice([50, 5000])
    start = rng.choice([0, 0, 3, 100, tlen - 1, tlen, tlen + 5]) if tlen < (1 << 20) else 0
    start = max(0, start)
    case["cov"] = {"start": start, "tlen": tlen, "runs": outcome(lambda: update_cov_vec(tlen, cigar, start))}
    case["pseudo_sym"] = outcome(lambda: gen_pesudo_maf_by_cigar(cigar, "", False)) if t_need < (1 << 13) else None
    if with_seqs and max(t_need, q_need, qp_need) < (1 << 13):
        # slices as long as the CIGAR consumes, and now and then longer or shorter (tails, insert_str / drain beyond the end)
        adj = rng.choice([0, 0, 0, 0, 5, -1, -7])
        t_seq = rand_seq(rng, max(0, t_need + adj))
        q_fwd = rand_seq(rng, max(0, q_need + rng.choice([0, 0, 0, adj])), dirty=rng.random() < 0.08)

        def insert():
            q = reverse_complement(q_fwd) if negative else q_fwd
            return list(parse_cigar_to_insert(cigar, t_seq, q))
        case["insert"] = {"t": t_seq, "q_fwd": q_fwd, "rows": outcome(insert)}
        qp_fwd = rand_seq(rng, max(0, qp_need + rng.choice([0, 0, 0, 4, -3])))

        def pseudo():
            q = reverse_complement(qp_fwd) if negative else qp_fwd
            return gen_pesudo_maf_by_cigar(cigar, q, True)
        case["pseudo_base"] = {"q_fwd": qp_fwd, "row": outcome(pseudo)}
    return case


def main():
    rng = random.Random(20260928)
    cases = []
    with open(os.path.join(ROOT, "tests", "golden", "testdotplot.paf")) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            cg = next(t for t in p[12:] if t.startswith("cg:Z:"))
            c = make_case(rng, cg, p[4] == "-")
            c["from"] = "testdotplot.paf"
            cases.append(c)
    for cg in adversarial_cigars(rng, 200):
        cases.append(make_case(rng, cg, rng.random() < 0.5))
    out = os.path.join(ROOT, "tests", "golden", "py_restatement.json")
    with open(out, "w") as f:
        json.dump({"about": "scripts/make_golden_py.py: a second, independent restatement of cigar.rs (Python, written from the "
                            "Rust text alone); checked against the C oracle and the HIP path by tests/test_golden_py.py",
                   "count_order": ["match", "mismatch", "ins_ev", "ins_bp", "del_ev", "del_bp", "inv_ins_ev", "inv_ins_bp",
                                   "inv_del_ev", "inv_del_bp", "inv_ev"],
                   "cases": cases}, f, ensure_ascii=True, separators=(",", ":"))
    kinds = {}
    for c in cases:
        k = "ok" if "ok" in c["stat"] else c["stat"].get("err", "panic")
        kinds[k] = kinds.get(k, 0) + 1
    print("wrote %s: %d cases, %d bytes; stat outcomes %s" % (out, len(cases), os.path.getsize(out), kinds))


if __name__ == "__main__":
    sys.exit(main())
