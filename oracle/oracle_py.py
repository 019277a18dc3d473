"""ctypes face of oracle/liboracle.so — the CPU checker.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by the wgatools_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("WGA_ORACLE_LIB") or os.path.join(ROOT, "oracle", "liboracle.so")  # the env override: sanitizer builds

COUNT_FIELDS = ("match", "mismatch", "ins_ev", "ins_bp", "del_ev", "del_bp", "inv_ins_ev",
                "inv_ins_bp", "inv_del_ev", "inv_del_bp", "inv_ev")


class Err(C.Structure):
    _fields_ = [("kind", C.c_int), ("arg", C.c_char * 64)]


class Counts(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in COUNT_FIELDS]

    def as_tuple(self):
        return tuple(int(getattr(self, k)) for k in COUNT_FIELDS)


class RecStat(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "aligned_size", "matched", "mismatched", "ins_event", "del_event", "ins_size", "del_size",
        "inv_ins_event", "inv_ins_size", "inv_del_event", "inv_del_size", "inv_event")] + [
        ("inv_size", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = [os.path.join(ROOT, "oracle", f) for f in ("oracle.c", "cpu_bench.c", "oracle.h")]
        if not os.environ.get("WGA_ORACLE_LIB") and (not os.path.exists(LIB_PATH)
                or any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH)
                       for s in src)):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        L = C.CDLL(LIB_PATH)
        P, Z, U = C.c_void_p, C.c_size_t, C.c_uint64
        L.orc_err_message.argtypes = [P, C.c_char_p, Z]
        L.orc_parse_paf_to_cigar.argtypes = [C.c_char_p, Z, C.c_int, P, P]
        L.orc_recstat_from.argtypes = [P, P]
        L.orc_reverse_complement.argtypes = [C.c_char_p, Z, C.c_char_p, P]
        L.orc_parse_cigar_to_insert.argtypes = [C.c_char_p, Z, P, P, P, P, P]
        L.orc_parse_maf_seq_to_cigar.argtypes = [C.c_char_p, Z, C.c_char_p, Z, C.c_int, P, P]
        L.orc_update_cov_vec.argtypes = [P, Z, C.c_char_p, Z, Z, P]
        L.orc_gen_pesudo_maf_by_cigar.argtypes = [C.c_char_p, Z, P, P, C.c_int, P]
        L.orc_call_var_maf_record.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, Z,
                                              U, U, U, U, U, C.c_int, C.c_int, C.c_int, U, Z, P, P]
        L.orc_call_within_var_paf.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, Z, C.c_char_p, Z,
                                              C.c_char_p, Z, U, U, U, U, C.c_int, C.c_int, U, P, P, P]
        L.orc_parse_cigar_to_trim.argtypes = [C.c_char_p, Z, P, P]
        L.orc_paf2chain_record.argtypes = [C.c_char_p, U, U, U, C.c_int, C.c_char_p, U, U, U, C.c_char_p, Z, U, P, P, P]
        L.orc_parse_maf_seq_to_trim.argtypes = [C.c_char_p, Z, C.c_char_p, Z, P]
        L.orc_parse_maf_seq_to_trim.restype = None
        L.orc_maf2chain_record.argtypes = [C.c_char_p, U, U, U, C.c_char_p, U, U, U, C.c_int, C.c_char_p, Z,
                                           C.c_char_p, Z, U, P, P]
        L.orc_maf2chain_record.restype = None
        L.orc_parse_chain_to_cigar.argtypes = [P, Z, C.c_int, P, P]
        L.orc_parse_chain_to_cigar.restype = None
        L.orc_parse_chain_to_insert.argtypes = [P, Z, P, P, P, P]
        L.orc_cigar_to_base_plotdata.argtypes = [C.c_char_p, Z, U, U, C.c_int, U, P, P, P]
        L.orc_maf_to_base_plotdata.argtypes = [C.c_char_p, Z, C.c_char_p, Z, U, U, C.c_int, U, P, P]
        L.orc_maf_to_base_plotdata.restype = None
        L.orc_tokenise.argtypes = [C.c_char_p, Z, P, P, Z, P, P, P, P]
        L.orc_ops_to_text.restype = Z
        L.orc_ops_to_text.argtypes = [P, Z, C.c_char_p, Z]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_cs_to_cigar.restype = C.c_void_p
        L.orc_cs_to_cigar.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_find_safe_chunk_boundary.restype = C.c_size_t
        L.orc_find_safe_chunk_boundary.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_size_t,
                                                   C.c_size_t, C.c_uint64]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, kind, arg, message):
        super().__init__(message)
        self.kind, self.arg, self.message = kind, arg, message


def _raise(err):
    buf = C.create_string_buffer(256)
    lib().orc_err_message(C.byref(err), buf, 256)
    raise OracleError(err.kind, err.arg.decode(errors="replace"), buf.value.decode(errors="replace"))


def parse_paf_to_cigar(cg, strand_neg):
    """cigar.rs:629-707 -> 11-tuple in wga_cigar_counts order"""
    cg = cg.encode() if isinstance(cg, str) else cg
    out, err = Counts(), Err()
    if lib().orc_parse_paf_to_cigar(cg, len(cg), int(strand_neg), C.byref(out), C.byref(err)):
        _raise(err)
    return out.as_tuple()


def recstat_from(counts):
    c = Counts(*counts)
    r = RecStat()
    lib().orc_recstat_from(C.byref(c), C.byref(r))
    return r


def reverse_complement(seq):
    seq = seq.encode() if isinstance(seq, str) else bytes(seq)
    out = C.create_string_buffer(len(seq) + 1)
    err = Err()
    if lib().orc_reverse_complement(seq, len(seq), out, C.byref(err)):
        _raise(err)
    return out.raw[: len(seq)]


def _malloc_copy(b):
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    p = libc.malloc(len(b) + 1)
    C.memmove(p, b, len(b))
    return C.c_void_p(p)


def parse_cigar_to_insert(cg, t_seq, q_seq):
    """cigar.rs:522-551: returns the two gapped rows (bytes)"""
    cg = cg.encode() if isinstance(cg, str) else cg
    t, q = _malloc_copy(bytes(t_seq)), _malloc_copy(bytes(q_seq))
    tn, qn = C.c_size_t(len(t_seq)), C.c_size_t(len(q_seq))
    err = Err()
    rc = lib().orc_parse_cigar_to_insert(cg, len(cg), C.byref(t), C.byref(tn), C.byref(q),
                                         C.byref(qn), C.byref(err))
    try:
        if rc:
            _raise(err)
        return C.string_at(t, tn.value), C.string_at(q, qn.value)
    finally:
        lib().orc_free(t)
        lib().orc_free(q)


def parse_maf_seq_to_cigar(t_row, q_row, strand_neg):
    """cigar.rs:344-432 -> (counts 11-tuple, cigar text)"""
    t_row, q_row = bytes(t_row), bytes(q_row)
    out = Counts()
    txt = C.c_void_p()
    lib().orc_parse_maf_seq_to_cigar(t_row, C.c_size_t(len(t_row)), q_row, C.c_size_t(len(q_row)),
                                     int(strand_neg), C.byref(out), C.byref(txt))
    s = C.string_at(txt).decode()
    lib().orc_free(txt)
    return out.as_tuple(), s


def update_cov_vec(cov, cg, start):
    """cigar.rs:710-741 on a numpy uint64 array (in place)"""
    cg = cg.encode() if isinstance(cg, str) else cg
    assert cov.dtype == np.uint64 and cov.flags.c_contiguous
    err = Err()
    if lib().orc_update_cov_vec(C.c_void_p(cov.ctypes.data), C.c_size_t(len(cov)), cg,
                                C.c_size_t(len(cg)), C.c_size_t(int(start)), C.byref(err)):
        _raise(err)


def gen_pesudo_maf_by_cigar(cg, q_seq, base):
    """cigar.rs:744-804 -> edited sequence (bytes)"""
    cg = cg.encode() if isinstance(cg, str) else cg
    q = _malloc_copy(bytes(q_seq))
    qn = C.c_size_t(len(q_seq))
    err = Err()
    rc = lib().orc_gen_pesudo_maf_by_cigar(cg, C.c_size_t(len(cg)), C.byref(q), C.byref(qn),
                                           int(base), C.byref(err))
    try:
        if rc:
            _raise(err)
        return C.string_at(q, qn.value)
    finally:
        lib().orc_free(q)


def cs_to_cigar(cs):
    cs = cs.encode() if isinstance(cs, str) else cs
    p = lib().orc_cs_to_cigar(cs, len(cs))
    s = C.string_at(p).decode()
    lib().orc_free(p)
    return s


def find_safe_chunk_boundary(t, q, start, chunk_size, svlen_cutoff):
    t, q = bytes(t), bytes(q)
    return lib().orc_find_safe_chunk_boundary(t, q, len(t), start, chunk_size, svlen_cutoff)


def call_var_maf_record(chro, q_chro, t_row, q_row, t_start, q_sline_start, q_sline_align, q_size,
                        strand_neg, if_snp, if_inv, svlen_cutoff, chunk_size=1000000):
    """caller.rs:115-149 + :388-608 for one MAF block -> VCF body text"""
    t_row, q_row = bytes(t_row), bytes(q_row)
    cols = min(len(t_row), len(q_row))
    out = C.c_void_p()
    out_len = C.c_size_t(0)
    rc = lib().orc_call_var_maf_record(
        chro.encode(), q_chro.encode(), t_row, q_row, C.c_size_t(cols), C.c_uint64(t_start),
        C.c_uint64(0), C.c_uint64(q_sline_start), C.c_uint64(q_sline_align), C.c_uint64(q_size),
        int(strand_neg), int(if_snp), int(if_inv), C.c_uint64(svlen_cutoff),
        C.c_size_t(chunk_size), C.byref(out), C.byref(out_len))
    s = C.string_at(out, out_len.value).decode() if out.value else ""
    if out.value:
        lib().orc_free(out)
    if rc:
        raise OracleError(rc, "", "call_within_var panicked")
    return s


def call_within_var(chro, q_chro, t_row, q_row, t_start, t_end, q_start, q_end, strand_neg, if_snp, svlen_cutoff,
                    if_inv):
    """caller.rs:388-608 on ONE chunk (no chunking, coordinates as given) -> VCF body text"""
    t_row, q_row = bytes(t_row), bytes(q_row)
    cols = min(len(t_row), len(q_row))
    out = C.c_void_p()
    out_len = C.c_size_t(0)
    L = lib()
    L.orc_call_within_var.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_uint64,
                                      C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                      C.c_void_p, C.c_void_p]
    rc = L.orc_call_within_var(chro.encode(), q_chro.encode(), t_row, q_row, cols, t_start, t_end, q_start, q_end,
                               int(strand_neg), int(if_snp), svlen_cutoff, int(if_inv), C.byref(out), C.byref(out_len))
    s = C.string_at(out, out_len.value).decode() if out.value else ""
    if out.value:
        L.orc_free(out)
    if rc:
        raise OracleError(rc, "", "call_within_var panicked")
    return s


def call_within_var_paf(chro, q_chro, cg, t_seq, q_seq, t_start, t_end, q_start, q_end, strand_neg,
                        if_snp, svlen_cutoff):
    """caller.rs:610-822 for one PAF record -> VCF body text"""
    cg = cg.encode() if isinstance(cg, str) else bytes(cg)
    t_seq, q_seq = bytes(t_seq), bytes(q_seq)
    out = C.c_void_p()
    out_len = C.c_size_t(0)
    err = Err()
    rc = lib().orc_call_within_var_paf(
        chro.encode(), q_chro.encode(), cg, C.c_size_t(len(cg)), t_seq, C.c_size_t(len(t_seq)), q_seq,
        C.c_size_t(len(q_seq)), C.c_uint64(t_start), C.c_uint64(t_end), C.c_uint64(q_start),
        C.c_uint64(q_end), int(strand_neg), int(if_snp), C.c_uint64(svlen_cutoff), C.byref(out),
        C.byref(out_len), C.byref(err))
    s = C.string_at(out, out_len.value).decode() if out.value else ""
    if out.value:
        lib().orc_free(out)
    if rc:
        _raise(err)
    return s


def parse_cigar_to_trim(cg):
    """cigar.rs:202-245 -> (head_ins, head_del, tail_ins, tail_del)"""
    cg = cg.encode() if isinstance(cg, str) else bytes(cg)
    out = (C.c_uint64 * 4)()
    err = Err()
    if lib().orc_parse_cigar_to_trim(cg, len(cg), out, C.byref(err)):
        _raise(err)
    return tuple(int(x) for x in out)


def paf2chain_record(q_name, q_size, q_start, q_end, strand_neg, t_name, t_size, t_start, t_end, cg, chain_id):
    """converter.rs:148-173 for one PAF record -> chain text (header, data lines, blank line)"""
    cg = cg.encode() if isinstance(cg, str) else bytes(cg)
    out = C.c_void_p()
    out_len = C.c_size_t(0)
    err = Err()
    rc = lib().orc_paf2chain_record(q_name.encode(), q_size, q_start, q_end, int(strand_neg), t_name.encode(),
                                    t_size, t_start, t_end, cg, len(cg), chain_id, C.byref(out),
                                    C.byref(out_len), C.byref(err))
    if rc:
        _raise(err)
    s = C.string_at(out, out_len.value)
    lib().orc_free(out)
    return s


def parse_maf_seq_to_trim(t_row, q_row):
    """cigar.rs:155-199 -> (head_ins, head_del, tail_ins, tail_del)"""
    t_row, q_row = bytes(t_row), bytes(q_row)
    out = (C.c_uint64 * 4)()
    lib().orc_parse_maf_seq_to_trim(t_row, len(t_row), q_row, len(q_row), out)
    return tuple(int(x) for x in out)


def maf2chain_record(t_name, t_size, t_start, t_align, q_name, q_size, q_start, q_align, strand_neg, t_row,
                     q_row, chain_id):
    """converter.rs:57-91 for one MAF block -> chain text (header, data lines, blank line)"""
    t_row, q_row = bytes(t_row), bytes(q_row)
    out = C.c_void_p()
    out_len = C.c_size_t(0)
    lib().orc_maf2chain_record(t_name.encode(), t_size, t_start, t_align, q_name.encode(), q_size, q_start,
                               q_align, int(strand_neg), t_row, len(t_row), q_row, len(q_row), chain_id,
                               C.byref(out), C.byref(out_len))
    s = C.string_at(out, out_len.value)
    lib().orc_free(out)
    return s


def _lines_array(lines):
    a = np.ascontiguousarray(np.asarray(lines, dtype=np.uint64).reshape(-1, 3))
    return a, C.c_void_p(a.ctypes.data)


def parse_chain_to_cigar(lines, strand_neg):
    """cigar.rs:554-627: lines = [(size, query_diff, target_diff)] -> (counts 11-tuple, CIGAR text)"""
    a, ptr = _lines_array(lines)
    out = Counts()
    txt = C.c_void_p()
    lib().orc_parse_chain_to_cigar(ptr, len(a), int(strand_neg), C.byref(out), C.byref(txt))
    s = C.string_at(txt).decode()
    lib().orc_free(txt)
    return out.as_tuple(), s


def parse_chain_to_insert(lines, t_seq, q_seq):
    """converter.rs:360-388: the two gapped rows (bytes); OracleError where insert_str panics"""
    a, ptr = _lines_array(lines)
    t, q = _malloc_copy(bytes(t_seq)), _malloc_copy(bytes(q_seq))
    tn, qn = C.c_size_t(len(t_seq)), C.c_size_t(len(q_seq))
    rc = lib().orc_parse_chain_to_insert(ptr, len(a), C.byref(t), C.byref(tn), C.byref(q), C.byref(qn))
    try:
        if rc:
            e = Err()
            e.kind = rc
            _raise(e)
        return C.string_at(t, tn.value), C.string_at(q, qn.value)
    finally:
        lib().orc_free(t)
        lib().orc_free(q)


def _take_segs(ptr, n):
    if not n:
        return np.zeros((0, 5), dtype=np.uint64)
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(n * 5,)).copy().reshape(n, 5)
    lib().orc_free(ptr)
    return a


def cigar_to_base_plotdata(cg, t_start, q_start, strand_neg, cutoff):
    """cigar.rs:917-952 -> (n, 5) uint64 rows ref_start, ref_end, query_start, query_end, kind"""
    cg = cg.encode() if isinstance(cg, str) else bytes(cg)
    segs, n, err = C.c_void_p(), C.c_size_t(0), Err()
    if lib().orc_cigar_to_base_plotdata(cg, len(cg), t_start, q_start, int(strand_neg), int(cutoff), C.byref(segs),
                                        C.byref(n), C.byref(err)):
        _raise(err)
    return _take_segs(segs, n.value)


def maf_to_base_plotdata(t_row, q_row, t_start, q_start, strand_neg, cutoff):
    """cigar.rs:955-985"""
    t_row, q_row = bytes(t_row), bytes(q_row)
    segs, n = C.c_void_p(), C.c_size_t(0)
    lib().orc_maf_to_base_plotdata(t_row, len(t_row), q_row, len(q_row), t_start, q_start, int(strand_neg),
                                   int(cutoff), C.byref(segs), C.byref(n))
    return _take_segs(segs, n.value)


def ops_to_text(ops):
    """packed u32 ops (numpy) -> b'cg:Z:...' (helper for large samples; not a reference function)"""
    ops = np.ascontiguousarray(ops, dtype=np.uint32)
    cap = 12 * len(ops) + 16
    buf = C.create_string_buffer(cap)
    k = lib().orc_ops_to_text(C.c_void_p(ops.ctypes.data), len(ops), buf, cap)
    return buf.raw[:k]


def tokenise(text):
    """cigar.rs:43-75 over the CIGAR behind "cg:Z:": ([(length, first byte of the op char)], error kind,
    (offset, length) of the token the error message quotes)"""
    text = text.encode() if isinstance(text, str) else bytes(text)
    cap = len(text) + 1
    lens = (C.c_uint64 * cap)()
    opc = (C.c_ubyte * cap)()
    n, eo, el = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    err = Err()
    kind = lib().orc_tokenise(text, len(text), lens, opc, cap, C.byref(n), C.byref(err), C.byref(eo), C.byref(el))
    return [(int(lens[i]), int(opc[i])) for i in range(n.value)], int(kind), (eo.value, el.value)


class BenchResult(C.Structure):
    _fields_ = [("seconds", C.c_double), ("ops", C.c_uint64), ("out_bytes", C.c_uint64), ("checksum", C.c_uint64)]


def bench_run(mode, threads, cg_blob, cg_off, ops, op_off, strand, t_pool, t_off, t_len, q_pool, q_off, q_len):
    """oracle/cpu_bench.c: mode 0 ref-faithful stat, 1 ref-faithful paf2maf, 2 optimised stat + paf2maf; numpy arrays in"""
    L = lib()
    P = C.c_void_p
    L.orc_bench_run.argtypes = [C.c_int, C.c_int, C.c_uint32] + [P] * 11 + [P]
    n = len(strand)
    res = BenchResult()
    ptr = lambda a: a.ctypes.data_as(P) if a is not None else None
    rc = L.orc_bench_run(mode, threads, n, ptr(cg_blob), ptr(cg_off), ptr(ops), ptr(op_off), ptr(strand), ptr(t_pool),
                         ptr(t_off), ptr(t_len), ptr(q_pool), ptr(q_off), ptr(q_len), C.byref(res))
    if rc:
        raise RuntimeError("cpu_bench: a record failed (rc %d)" % rc)
    return res
