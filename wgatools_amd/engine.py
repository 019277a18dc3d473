"""Thin object layer over the C-ABI: device arrays and one method per entry point.

No computation happens here — every method forwards to libwgahip.so.  Arrays live in HBM
(`DeviceArray`); `upload` / `numpy()` are the only host<->device copies.
"""
import ctypes as C

import numpy as np

from . import _lib

COUNTS_DTYPE = np.dtype([(k, "<u8") for k in (
    "match", "mismatch", "ins_ev", "ins_bp", "del_ev", "del_bp", "inv_ins_ev", "inv_ins_bp",
    "inv_del_ev", "inv_del_bp", "inv_ev")])
DIAG_DTYPE = np.dtype([("bad_op_idx", "<u8"), ("panic_op_idx", "<u8"), ("bad_base_pos", "<u8")])
CHAIN_TRIM_DTYPE = np.dtype([(k, "<u8") for k in ("head_ins", "head_del", "tail_ins", "tail_del")])
TOK_ERR_DTYPE = np.dtype([("err", np.int32), ("tok_len", np.uint32), ("tok_off", np.uint64)])
CLASS_SUMS_DTYPE = np.dtype([(k, "<u8") for k in ("mx", "i", "d", "s", "o")])
FA_CONTIG_DTYPE = np.dtype([(k, "<u8") for k in ("hdr_start", "hdr_end", "pool_off", "len")])
NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
MAF_LINE_DTYPE = np.dtype([("num", np.uint64, 3), ("name_off", np.uint64), ("seq_off", np.uint64),
                           ("seq_len", np.uint64), ("name_len", np.uint32), ("strand_neg", np.uint8),
                           ("status", np.uint8), ("pad", np.uint8, 2)])
PAF_LINE_DTYPE = np.dtype([("num", np.uint64, 9), ("qname_off", np.uint64), ("tname_off", np.uint64),
                           ("cg_beg", np.uint64), ("cg_end", np.uint64), ("qname_len", np.uint32),
                           ("tname_len", np.uint32), ("n_fields", np.uint32), ("strand_neg", np.uint8),
                           ("status", np.uint8), ("pad", np.uint8, 2)])

OP_CODES = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
OP_I_CONT, OP_D_CONT, OP_OTHER = 9, 10, 11
OP_MAX_LEN = (1 << 28) - 1

REC_ERR_NAMES = {0: "ok", 1: "CigarTagNotFound", 2: "CigarOpInvalid", 3: "ParseIntError",
                 4: "InvalidBase", 5: "NomErr", 6: "panic"}


class DeviceArray:
    """A typed view of an HBM allocation owned by (or borrowed into) an Engine."""

    def __init__(self, eng, ptr, shape, dtype, owner=True):
        self.eng = eng
        self.ptr = ptr
        self.shape = tuple(shape) if not np.isscalar(shape) else (int(shape),)
        self.dtype = np.dtype(dtype)
        self.owner = owner

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        self.eng._check(self.eng.lib.wga_memcpy_d2h(self.eng.ctx, out.ctypes.data, self.ptr,
                                                    self.nbytes))
        return out

    def fill(self, byte):
        self.eng._check(self.eng.lib.wga_memset(self.eng.ctx, self.ptr, byte, self.nbytes))
        return self

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (int(self.ptr), False), "version": 2,
                "strides": None}

    def torch(self, device):
        """zero-copy torch view of this allocation (plumbing for the harness: torch only sees the bytes); the
        DeviceArray must outlive the tensor"""
        import torch
        t = torch.as_tensor(self, device=device)
        if t.data_ptr() != int(self.ptr):
            raise RuntimeError("torch copied the buffer instead of viewing it")
        return t

    def free(self):
        if self.owner and self.ptr:
            self.eng.lib.wga_free(self.eng.ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _p(x):
    """device pointer of a DeviceArray / torch tensor / int / None"""
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class Batch:
    """A device-resident CSR batch of packed CIGARs (wga_cigar_batch)."""

    def __init__(self, ops, op_off, strand_neg, n, n_ops):
        self.ops, self.op_off, self.strand_neg = ops, op_off, strand_neg
        self.n, self.n_ops = int(n), int(n_ops)
        self.c = _lib.CigarBatch(_p(ops), _p(op_off), _p(strand_neg), self.n_ops, self.n)


VCF_ERR_DTYPE = np.dtype([("item", "<u8"), ("kind", "<u4"), ("ch", "<u4")])
MAF_VCF_REC_DTYPE = np.dtype([("t_name_off", "<u8"), ("q_name_off", "<u8"), ("t_name_len", "<u4"), ("q_name_len", "<u4"),
                              ("t_start", "<u8"), ("q_start", "<u8"), ("q_size", "<u8"), ("q_neg", "<u4"), ("pad", "<u4")])


class Engine:
    def __init__(self, device=0, lib=None):
        self.lib = lib if lib is not None else _lib.load()
        ctx = C.c_void_p()
        rc = self.lib.wga_ctx_create(device, C.byref(ctx))
        if rc != 0:
            raise _lib.WgaError("wga_ctx_create failed (%d): %s" % (
                rc, self.lib.wga_last_error().decode()))
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.wga_ctx_destroy(self.ctx)
            self.ctx = None

    def _check(self, rc):
        if rc != 0:
            raise _lib.WgaError("libwgahip call failed (%d): %s" % (
                rc, self.lib.wga_last_error().decode()))

    # ---- memory -----------------------------------------------------------------------------
    def empty(self, shape, dtype):
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if not np.isscalar(shape) else int(shape)
        ptr = C.c_void_p()
        self._check(self.lib.wga_malloc(self.ctx, max(n * dt.itemsize, 16), C.byref(ptr)))
        return DeviceArray(self, ptr.value, shape, dt)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        d = self.empty(arr.shape, arr.dtype)
        self._check(self.lib.wga_memcpy_h2d(self.ctx, d.ptr, arr.ctypes.data, arr.nbytes))
        self.sync()  # the host array may be a temporary
        return d

    def copy_into(self, dst, arr):
        """host array -> an existing device array (the same pointer keeps its address)"""
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.wga_memcpy_h2d(self.ctx, dst.ptr, arr.ctypes.data, arr.nbytes))
        self.sync()

    def sync(self):
        self._check(self.lib.wga_sync(self.ctx))

    def set_stream(self, hip_stream):
        """launch on this hipStream_t (0 / None = HIP's default stream)"""
        self._check(self.lib.wga_ctx_set_stream(self.ctx, hip_stream or None))

    def reset_stream(self):
        self._check(self.lib.wga_ctx_reset_stream(self.ctx))

    def set_param(self, name, value):
        self._check(self.lib.wga_ctx_set_param(self.ctx, name.encode(), int(value)))

    def get_param(self, name):
        v = C.c_int64(0)
        self._check(self.lib.wga_ctx_get_param(self.ctx, name.encode(), C.byref(v)))
        return v.value

    def expand_timing(self):
        """(summed ms, launches) of the expand kernel proper since the last call ("expand_timing" param)"""
        ms, n = C.c_double(0.0), C.c_uint32(0)
        self._check(self.lib.wga_ctx_expand_timing(self.ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- host packer ------------------------------------------------------------------------
    def pack_cigar(self, text):
        """CIGAR text (after 'cg:Z:') -> (ops u32[], err code, (tok_off, tok_len))."""
        if isinstance(text, str):
            text = text.encode()
        n = C.c_size_t()
        err = C.c_int32()
        eo, el = C.c_size_t(), C.c_size_t()
        cap = max(1, text.count(b"M") + len(text))  # generous first guess
        ops = np.empty(cap, dtype=np.uint32)
        rc = self.lib.wga_cigar_pack(text, len(text), ops.ctypes.data, cap, C.byref(n),
                                     C.byref(err), C.byref(eo), C.byref(el))
        if rc == -5:  # WGA_E_TOO_SMALL
            cap = n.value
            ops = np.empty(cap, dtype=np.uint32)
            rc = self.lib.wga_cigar_pack(text, len(text), ops.ctypes.data, cap, C.byref(n),
                                         C.byref(err), C.byref(eo), C.byref(el))
        self._check(rc)
        return ops[: n.value].copy(), err.value, (eo.value, el.value)

    def make_batch(self, ops, op_off, strand_neg):
        """numpy CSR arrays -> device Batch"""
        ops = np.ascontiguousarray(ops, dtype=np.uint32)
        op_off = np.ascontiguousarray(op_off, dtype=np.uint64)
        strand_neg = np.ascontiguousarray(strand_neg, dtype=np.uint8)
        n = len(strand_neg)
        assert len(op_off) == n + 1 and int(op_off[-1]) == len(ops)
        return Batch(self.upload(ops), self.upload(op_off), self.upload(strand_neg), n, len(ops))

    def make_batch_device(self, ops, op_off, strand_neg, n, n_ops):
        """device arrays (e.g. written by the tokeniser or the K11 bridges) -> Batch"""
        return Batch(ops, op_off, strand_neg, int(n), int(n_ops))

    # ---- kernels ----------------------------------------------------------------------------
    def tile_ws(self, n_ops):
        return self.empty(self.lib.wga_tile_ws_bytes(int(n_ops)), np.uint8)

    def cigar_stat(self, batch, counts=None, diag=None, tile_ws=None, want_tiles=True):
        counts = counts if counts is not None else self.empty(batch.n, COUNTS_DTYPE)
        diag = diag if diag is not None else self.empty(batch.n, DIAG_DTYPE)
        if tile_ws is None and want_tiles:
            tile_ws = self.tile_ws(batch.n_ops)
        self._check(self.lib.wga_cigar_stat(self.ctx, C.byref(batch.c), _p(counts), _p(diag),
                                            _p(tile_ws)))
        return counts, diag, tile_ws

    def cigar_class_sums(self, batch, sums=None):
        sums = sums if sums is not None else self.empty(batch.n, CLASS_SUMS_DTYPE)
        self._check(self.lib.wga_cigar_class_sums(self.ctx, C.byref(batch.c), _p(sums)))
        return sums

    def paf2maf_layout(self, n, counts, t_src_len, q_src_len, pre_t=None, pre_q=None, post=None,
                       t_row_off=None, q_row_off=None, rec_off=None):
        t_row_off = t_row_off if t_row_off is not None else self.empty(n, np.uint64)
        q_row_off = q_row_off if q_row_off is not None else self.empty(n, np.uint64)
        rec_off = rec_off if rec_off is not None else self.empty(n + 1, np.uint64)
        self._check(self.lib.wga_paf2maf_layout(self.ctx, n, _p(counts), _p(t_src_len),
                                                _p(q_src_len), _p(pre_t), _p(pre_q), _p(post),
                                                _p(t_row_off), _p(q_row_off), _p(rec_off)))
        return t_row_off, q_row_off, rec_off

    def paf2maf_expand(self, batch, counts, tile_ws, t_fa, t_fa_bytes, t_src_off, t_src_len, q_fa,
                       q_fa_bytes, q_src_off, q_src_len, out, t_row_off, q_row_off, diag):
        self._check(self.lib.wga_paf2maf_expand(
            self.ctx, C.byref(batch.c), _p(counts), _p(tile_ws), _p(t_fa), int(t_fa_bytes),
            _p(t_src_off), _p(t_src_len), _p(q_fa), int(q_fa_bytes), _p(q_src_off), _p(q_src_len),
            _p(out), _p(t_row_off), _p(q_row_off), _p(diag)))

    def bgzf_inflate(self, d_in, in_bytes, n_blocks, blocks, out, status):
        """BGZF members inflated on the device (wga_bgzf_inflate); blocks: n x (in_off u64, in_len u32, out_len u32, out_off u64)"""
        self._check(self.lib.wga_bgzf_inflate(self.ctx, _p(d_in), int(in_bytes), int(n_blocks), _p(blocks), _p(out), _p(status)))

    def bgzf_compress(self, d_in, n_bytes, out=None, eof_marker=True, in_offset=0, out_offset=0, out_cap=None):
        """bytes in HBM -> BGZF members (wga_bgzf_compress, K18); returns (DeviceArray of the worst-case size, bytes used).
        in_offset / out_offset: byte offsets into d_in / out (any alignment).  The capacity handed to the library is what `out`
        really holds behind out_offset (a DeviceArray's or a tensor's own size; out_cap for a raw pointer): the library's
        own check then refuses a buffer that is too small instead of writing past it."""
        cap = int(self.lib.wga_bgzf_bound(int(n_bytes)))
        if out is None:
            out = self.empty(cap + int(out_offset), np.uint8)
        if out_cap is None:
            if isinstance(out, DeviceArray):
                out_cap = out.nbytes - int(out_offset)
            elif hasattr(out, "numel") and hasattr(out, "element_size"):
                out_cap = out.numel() * out.element_size() - int(out_offset)
            else:
                raise ValueError("bgzf_compress: out_cap is required when `out` is a raw pointer")
        if out_cap < 0:
            raise ValueError("bgzf_compress: out_offset lies behind the end of `out`")
        used = C.c_uint64(0)
        src = _p(d_in)
        self._check(self.lib.wga_bgzf_compress(self.ctx, (src + int(in_offset)) if src else None, int(n_bytes),
                                               _p(out) + int(out_offset), int(out_cap), C.byref(used), 1 if eof_marker else 0))
        return out, int(used.value)

    def scatter_bytes(self, n, src, src_off, dst, dst_off):
        self._check(self.lib.wga_scatter_bytes(self.ctx, n, _p(src), _p(src_off), _p(dst),
                                               _p(dst_off)))

    def exclusive_scan_u64(self, n, d_in, d_out=None):
        d_out = d_out if d_out is not None else self.empty(n + 1, np.uint64)
        self._check(self.lib.wga_exclusive_scan_u64(self.ctx, n, _p(d_in), _p(d_out)))
        return d_out

    def maf_pair_stat(self, n, rows, t_off, q_off, cols, strand_neg, counts=None, run_cnt=None,
                      runs=None, run_off=None):
        counts = counts if counts is not None else self.empty(n, COUNTS_DTYPE)
        run_cnt = run_cnt if run_cnt is not None else self.empty(n, np.uint64)
        self._check(self.lib.wga_maf_pair_stat(self.ctx, n, _p(rows), _p(t_off), _p(q_off),
                                               _p(cols), _p(strand_neg), _p(counts), _p(run_cnt),
                                               _p(runs), _p(run_off)))
        return counts, run_cnt

    def maf_call_runs(self, n, rows, t_off, q_off, cols, run_cnt=None, runs=None, run_off=None):
        run_cnt = run_cnt if run_cnt is not None else self.empty(n, np.uint64)
        self._check(self.lib.wga_maf_call_runs(self.ctx, n, _p(rows), _p(t_off), _p(q_off), _p(cols),
                                               _p(run_cnt), _p(runs), _p(run_off)))
        return run_cnt

    def maf_call_vcf(self, n, rows, t_off, q_off, cols, runs, run_off, recs, names, snp, inv, svlen, chunk_size,
                     nbytes=None, err=None, out=None, out_off=None):
        """the rules and VCF rows of `call` on MAF (wga_maf_call_vcf, K19) on K4's run list: count pass when out is None
        (returns nbytes, err), fill pass otherwise.  recs: n x MAF_VCF_REC_DTYPE"""
        if out is None:
            nbytes = nbytes if nbytes is not None else self.empty(n, np.uint64)
            err = err if err is not None else self.empty(n, VCF_ERR_DTYPE)
        self._check(self.lib.wga_maf_call_vcf(self.ctx, n, _p(rows), _p(t_off), _p(q_off), _p(cols), _p(runs), _p(run_off),
                                              _p(recs), _p(names), 1 if snp else 0, 1 if inv else 0, int(svlen), int(chunk_size),
                                              _p(nbytes) if out is None else None, _p(err) if out is None else None,
                                              _p(out), _p(out_off)))
        return nbytes, err

    def cigar_tokenise(self, n, text, text_off, op_cnt=None, err=None, ops=None, op_off=None):
        """device tokeniser (wga_cigar_tokenise): count pass when ops is None, fill pass otherwise"""
        op_cnt = op_cnt if op_cnt is not None else self.empty(n, np.uint64)
        err = err if err is not None else self.empty(n, TOK_ERR_DTYPE)
        self._check(self.lib.wga_cigar_tokenise(self.ctx, n, _p(text), _p(text_off), _p(op_cnt), _p(err),
                                                _p(ops), _p(op_off)))
        return op_cnt, err

    def cigar_chain(self, batch, trim=None, nbytes=None, diag=None, out=None, out_off=None):
        """paf2chain data lines: count pass when out is None (trim, nbytes, diag), fill pass otherwise"""
        if out is None:
            trim = trim if trim is not None else self.empty(batch.n, CHAIN_TRIM_DTYPE)
            nbytes = nbytes if nbytes is not None else self.empty(batch.n, np.uint64)
            diag = diag if diag is not None else self.empty(batch.n, DIAG_DTYPE)
        self._check(self.lib.wga_cigar_chain(self.ctx, C.byref(batch.c), _p(trim), _p(nbytes), _p(diag),
                                             _p(out), _p(out_off)))
        return trim, nbytes, diag

    def maf_runs_ops(self, n, n_elems, runs, run_off, cols, cnt=None, out=None, out_off=None):
        """K3 runs -> packed ops (count pass when out is None)"""
        cnt = cnt if cnt is not None or out is not None else self.empty(n, np.uint64)
        self._check(self.lib.wga_maf_runs_ops(self.ctx, n, int(n_elems), _p(runs), _p(run_off), _p(cols),
                                              _p(cnt), _p(out), _p(out_off)))
        return cnt

    def maf_runs_cigar_text(self, n, n_elems, runs, run_off, cols, cnt=None, out=None, out_off=None):
        """K3 runs -> maf2paf's cg:Z: text (count pass when out is None)"""
        cnt = cnt if cnt is not None or out is not None else self.empty(n, np.uint64)
        self._check(self.lib.wga_maf_runs_cigar_text(self.ctx, n, int(n_elems), _p(runs), _p(run_off),
                                                     _p(cols), _p(cnt), _p(out), _p(out_off)))
        return cnt

    def chain_lines_ops(self, n, n_elems, lines, line_off, cnt=None, out=None, out_off=None):
        """chain data lines (size, D, I) -> packed ops (count pass when out is None)"""
        cnt = cnt if cnt is not None or out is not None else self.empty(n, np.uint64)
        self._check(self.lib.wga_chain_lines_ops(self.ctx, n, int(n_elems), _p(lines), _p(line_off), _p(cnt),
                                                 _p(out), _p(out_off)))
        return cnt

    def chain_lines_cigar_text(self, n, n_elems, lines, line_off, cnt=None, out=None, out_off=None):
        """chain data lines -> chain2paf's CIGAR text (count pass when out is None)"""
        cnt = cnt if cnt is not None or out is not None else self.empty(n, np.uint64)
        self._check(self.lib.wga_chain_lines_cigar_text(self.ctx, n, int(n_elems), _p(lines), _p(line_off),
                                                        _p(cnt), _p(out), _p(out_off)))
        return cnt

    def cigar_dotplot(self, batch, cutoff, t_start, q_start, seg_cnt=None, segs=None, seg_off=None):
        """dotplot base-level segments (5 u64 each): count pass when segs is None"""
        seg_cnt = seg_cnt if seg_cnt is not None or segs is not None else self.empty(batch.n, np.uint64)
        self._check(self.lib.wga_cigar_dotplot(self.ctx, C.byref(batch.c), int(cutoff), _p(t_start), _p(q_start),
                                               _p(seg_cnt), _p(segs), _p(seg_off)))
        return seg_cnt

    def paf_split(self, text, n_bytes, lines=None):
        """K13: number of text lines (lines is None), or the wga_paf_line of every line"""
        nl = C.c_uint64(0)
        cap = 0 if lines is None else (lines.numel() if hasattr(lines, "numel") else lines.size)
        self._check(self.lib.wga_paf_split(self.ctx, _p(text), int(n_bytes), C.byref(nl), _p(lines), int(cap)))
        return int(nl.value)

    def maf_split(self, text, n_bytes, lines=None):
        """K14: number of text lines (lines is None), or the wga_maf_line of every line"""
        nl = C.c_uint64(0)
        cap = 0 if lines is None else (lines.numel() if hasattr(lines, "numel") else lines.size)
        self._check(self.lib.wga_maf_split(self.ctx, _p(text), int(n_bytes), C.byref(nl), _p(lines), int(cap)))
        return int(nl.value)

    def cigar_tokenise_spans(self, n, text, beg, end, op_cnt=None, err=None, ops=None, op_off=None):
        """device tokeniser on spans text[beg[i], end[i]) (e.g. the cg:Z: texts inside a PAF file)"""
        op_cnt = op_cnt if op_cnt is not None else self.empty(n, np.uint64)
        err = err if err is not None else self.empty(n, TOK_ERR_DTYPE)
        self._check(self.lib.wga_cigar_tokenise_spans(self.ctx, n, _p(text), _p(beg), _p(end), _p(op_cnt), _p(err),
                                                      _p(ops), _p(op_off)))
        return op_cnt, err

    def counts_total(self, n, counts, totals=None):
        """column sums of the n x 11 counter matrix (stat totals)"""
        totals = totals if totals is not None else self.empty(11, np.uint64)
        self._check(self.lib.wga_counts_total(self.ctx, int(n), _p(counts), _p(totals)))
        return totals

    def fasta_pool(self, text, n_bytes):
        """FASTA text in HBM -> (pool DeviceArray, contig table as numpy FA_CONTIG_DTYPE); two calls of wga_fasta_pool"""
        nc, nb = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.wga_fasta_pool(self.ctx, _p(text), int(n_bytes), C.byref(nc), C.byref(nb), None, None))
        pool = self.empty(max(1, nb.value), np.uint8)
        contigs = self.empty(max(1, nc.value), FA_CONTIG_DTYPE)
        self._check(self.lib.wga_fasta_pool(self.ctx, _p(text), int(n_bytes), C.byref(nc), C.byref(nb), _p(pool), _p(contigs)))
        pool.shape = (nb.value,)
        return pool, contigs.numpy()[: nc.value]

    def paf_call_events(self, batch, svlen, snp, ev_cnt=None, ev=None, ev_off=None):
        ev_cnt = ev_cnt if ev_cnt is not None else self.empty(batch.n, np.uint64)
        self._check(self.lib.wga_paf_call_events(self.ctx, C.byref(batch.c), int(svlen), int(bool(snp)),
                                                 _p(ev_cnt), _p(ev), _p(ev_off)))
        return ev_cnt

    def pafcov_accumulate(self, batch, target_id, t_start, cov_off, cov_len, cov, total_cov):
        self._check(self.lib.wga_pafcov_accumulate(self.ctx, C.byref(batch.c), _p(target_id),
                                                   _p(t_start), _p(cov_off), _p(cov_len), _p(cov),
                                                   int(total_cov)))

    def pafcov_accumulate_final(self, batch, target_id, t_start, cov_off, cov_len, n_targets, cov, total_cov):
        """the last batch's marks and the marks -> counts scan in one pass over the array"""
        self._check(self.lib.wga_pafcov_accumulate_final(self.ctx, C.byref(batch.c), _p(target_id), _p(t_start), _p(cov_off),
                                                         _p(cov_len), int(n_targets), _p(cov), int(total_cov)))

    def pafcov_finalize(self, n_targets, cov_off, cov_len, cov):
        self._check(self.lib.wga_pafcov_finalize(self.ctx, n_targets, _p(cov_off), _p(cov_len),
                                                 _p(cov)))

    def pafcov_format(self, name, cov, p0, count, line_off=None, out=None):
        """BED text of pafcov for positions p0 .. p0+count-1 (cov = device pointer of position p0's counter)"""
        line_off = line_off if line_off is not None else self.empty(count + 1, np.uint64)
        self._check(self.lib.wga_pafcov_format(self.ctx, _p(name), int(name.numel() if hasattr(name, 'numel') else name.size), _p(cov), int(p0), int(count),
                                               _p(line_off), _p(out)))
        return line_off

    def pafpseudo_fill(self, batch, base_mode, q_fa, q_fa_bytes, q_src_off, q_src_len, skip, out,
                       dst_off, diag=None):
        diag = diag if diag is not None else self.empty(batch.n, DIAG_DTYPE)
        self._check(self.lib.wga_pafpseudo_fill(self.ctx, C.byref(batch.c), int(base_mode),
                                                _p(q_fa), int(q_fa_bytes), _p(q_src_off),
                                                _p(q_src_len), _p(skip), _p(out), _p(dst_off),
                                                _p(diag)))
        return diag
