"""ctypes binding of the C-ABI in include/wga_hip.h.

The product library is wgatools_amd/libwgahip.so (HIP, gfx950).  There is no CPU fallback: if the
library is missing or no GPU is visible the calls raise.  (tests/ may pass the path of the SIMT
emulator build to `load()` to exercise kernel logic on a CPU; the product never does.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libwgahip.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p


class CigarBatch(C.Structure):
    """wga_cigar_batch"""
    _fields_ = [
        ("d_ops", vp),
        ("d_op_off", vp),
        ("d_strand_neg", vp),
        ("n_ops", C.c_uint64),
        ("n", C.c_uint32),
    ]


# name -> (restype, argtypes); every symbol include/wga_hip.h declares
PROTOTYPES = {
    "wga_abi_version": (C.c_int, []),
    "wga_last_error": (C.c_char_p, []),
    "wga_device_count": (C.c_int, []),
    "wga_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "wga_ctx_destroy": (None, [vp]),
    "wga_ctx_set_stream": (C.c_int, [vp, vp]),
    "wga_ctx_reset_stream": (C.c_int, [vp]),
    "wga_ctx_set_param": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "wga_ctx_get_param": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "wga_ctx_expand_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "wga_sync": (C.c_int, [vp]),
    "wga_malloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "wga_free": (C.c_int, [vp, vp]),
    "wga_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "wga_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "wga_memset": (C.c_int, [vp, vp, C.c_int, C.c_size_t]),
    "wga_cigar_pack": (C.c_int, [C.c_char_p, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_size_t),
                                 C.POINTER(C.c_size_t)]),
    "wga_cigar_pack_bound": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "wga_tile_ws_bytes": (C.c_size_t, [C.c_uint64]),
    "wga_cigar_stat": (C.c_int, [vp, C.POINTER(CigarBatch), vp, vp, vp]),
    "wga_cigar_class_sums": (C.c_int, [vp, C.POINTER(CigarBatch), vp]),
    "wga_paf2maf_layout": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "wga_paf2maf_expand": (C.c_int, [vp, C.POINTER(CigarBatch), vp, vp, vp, C.c_uint64, vp, vp,
                                     vp, C.c_uint64, vp, vp, vp, vp, vp, vp]),
    "wga_scatter_bytes": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp]),
    "wga_maf_pair_stat": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "wga_maf_call_runs": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]),
    "wga_cigar_tokenise": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp]),
    "wga_cigar_chain": (C.c_int, [vp, C.POINTER(CigarBatch), vp, vp, vp, vp, vp]),
    "wga_maf_runs_ops": (C.c_int, [vp, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp]),
    "wga_maf_runs_cigar_text": (C.c_int, [vp, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp]),
    "wga_chain_lines_ops": (C.c_int, [vp, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp]),
    "wga_chain_lines_cigar_text": (C.c_int, [vp, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp]),
    "wga_cigar_dotplot": (C.c_int, [vp, C.POINTER(CigarBatch), C.c_uint64, vp, vp, vp, vp, vp]),
    "wga_paf_split": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(C.c_uint64), vp, C.c_uint64]),
    "wga_maf_split": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(C.c_uint64), vp, C.c_uint64]),
    "wga_cigar_tokenise_spans": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]),
    "wga_counts_total": (C.c_int, [vp, C.c_uint32, vp, vp]),
    "wga_host_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(C.c_void_p)]),
    "wga_host_free": (C.c_int, [vp, vp]),
    "wga_memcpy_d2h_async": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "wga_fasta_pool": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp, vp]),
    "wga_paf_call_events": (C.c_int, [vp, C.POINTER(CigarBatch), C.c_uint64, C.c_int, vp, vp, vp]),
    "wga_bgzf_inflate": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp]),
    "wga_bgzf_bound": (C.c_uint64, [C.c_uint64]),
    "wga_bgzf_compress": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]),
    "wga_paf_call_vcf": (C.c_int, [vp, C.POINTER(CigarBatch), C.c_uint64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "wga_maf_call_vcf": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                   vp, vp, vp, vp]),
    "wga_pafcov_accumulate": (C.c_int, [vp, C.POINTER(CigarBatch), vp, vp, vp, vp, vp, C.c_uint64]),
    "wga_pafcov_accumulate_final": (C.c_int, [vp, C.POINTER(CigarBatch), vp, vp, vp, vp, C.c_uint32, vp, C.c_uint64]),
    "wga_pafcov_format": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "wga_pafcov_finalize": (C.c_int, [vp, C.c_uint32, vp, vp, vp]),
    "wga_pafpseudo_fill": (C.c_int, [vp, C.POINTER(CigarBatch), C.c_int, vp, C.c_uint64, vp, vp,
                                     vp, vp, vp, vp]),
    "wga_reduce_scatter_i32": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp), C.c_uint64]),
    "wga_exclusive_scan_u64": (C.c_int, [vp, C.c_uint32, vp, vp]),
}


class WgaError(RuntimeError):
    pass


def load(path=None, require_all=True):
    """dlopen the engine and attach prototypes.  Raises if the library or a symbol is missing."""
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise WgaError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and require_all:
        raise WgaError("%s lacks symbols: %s" % (path, ", ".join(missing)))
    lib._wga_path = path
    return lib
