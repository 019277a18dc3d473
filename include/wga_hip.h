/*
 * wga_hip.h — C-ABI of libwgahip.so: the MI355X (gfx950) engine for wgatools' CIGAR-driven hot path.
 *
 * Every compute entry point replaces one per-record Rust function of the reference (cited per
 * function as /root/reference-relative file:line) with a *batched* call over many records.
 *
 * Conventions
 *   - plain C, no ownership transfer: the caller allocates every input and output buffer.
 *   - every `const T* d_*` / `T* d_*` argument is a DEVICE pointer (HBM resident).  Callers that
 *     do not link HIP themselves (a Rust/cgo/ctypes binding) use wga_malloc/wga_memcpy_* below.
 *   - `d_ops` must be 16-byte aligned (hipMalloc / wga_malloc guarantee 256).
 *   - calls are asynchronous on the context's stream; wga_sync() (or a D2H copy) waits.
 *   - return value: 0 = WGA_OK, negative = wga_status.  Per-record problems never fail the call:
 *     they are reported in `wga_rec_diag` so the host can print the reference's message for the
 *     first failing record in input order (reference: errors.rs:45-74, main.rs:14-21).
 *
 * Packed CIGAR op (u32): len << 4 | code, len < 2^28.  Codes follow BAM for the nine standard
 * ops; the packer (wga_cigar_pack) splits longer lengths into several ops and marks the pieces of
 * a split I / D with continuation codes so that event counts stay exact.
 */
#ifndef WGA_HIP_H
#define WGA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: what a count call leaves in the context for its fill call (wga_maf_runs_* / wga_chain_lines_*, the op walks
 *    wga_paf_call_events / wga_cigar_chain / wga_cigar_dotplot over long records, wga_cigar_class_sums -> wga_pafpseudo_fill)
 *    is ONE-SHOT — the first fill call that takes it consumes it — and is dropped when one of the arrays it was made from is
 *    written through wga_memcpy_h2d / wga_memset or freed; "expand_variant" takes -1, 0, 2, 3.
 * 3: the output-placement calls (wga_paf2maf_expand_place, wga_arena_alloc, wga_arena_probe) and the parameters
 *    "expand_autotune" / "expand_alias" are gone — the streaming row kernel made them unnecessary —; new:
 *    wga_pafcov_accumulate_final; wga_reduce_scatter_i32 no longer waits on the host (its result is ordered on the contexts'
 *    streams) and wga_pafcov_finalize refuses overlapping target ranges.  No struct layout changed. */
#define WGA_ABI_VERSION 3

/* ---- status codes (call level) ------------------------------------------------------------ */
enum wga_status {
  WGA_OK = 0,
  WGA_E_INVALID_ARG = -1,
  WGA_E_NO_DEVICE = -2,
  WGA_E_HIP = -3, /* a HIP runtime call failed; see wga_last_error() */
  WGA_E_OOM = -4,
  WGA_E_TOO_SMALL = -5 /* caller buffer too small (host packer) */
};

/* ---- packed op codes ---------------------------------------------------------------------- */
enum wga_op_code {
  WGA_OP_M = 0,
  WGA_OP_I = 1,
  WGA_OP_D = 2,
  WGA_OP_N = 3,
  WGA_OP_S = 4,
  WGA_OP_H = 5,
  WGA_OP_P = 6,
  WGA_OP_EQ = 7,
  WGA_OP_X = 8,
  WGA_OP_I_CONT = 9,   /* continuation of a split I: same bases, no new event */
  WGA_OP_D_CONT = 10,  /* continuation of a split D */
  WGA_OP_OTHER = 11    /* any other single-char op (`B`, `z`, `é`, ...): the reference accepts the
                          token and lets each consumer decide (cigar.rs:43-56) */
};
#define WGA_OP_LEN_BITS 28
#define WGA_OP_MAX_LEN ((1u << WGA_OP_LEN_BITS) - 1u)
#define WGA_PACK_OP(len, code) (((uint32_t)(len) << 4) | (uint32_t)(code))

/* ---- per-record error codes: 1:1 with the WGAError variants the hot path can raise --------- */
enum wga_rec_err {
  WGA_REC_OK = 0,
  WGA_REC_CIGAR_TAG_NOT_FOUND = 1, /* errors.rs:57  "CIGAR start tag not found"            */
  WGA_REC_CIGAR_OP_INVALID = 2,    /* errors.rs:59  "CIGAR OP `{0}` invalid"               */
  WGA_REC_PARSE_INT = 3,           /* errors.rs:51  "Parse `{0}` Into Integer Error"       */
  WGA_REC_INVALID_BASE = 4,        /* errors.rs:73  "Invalid Base: `{0}`"                  */
  WGA_REC_NOM = 5,                 /* errors.rs:45  nom tag error                          */
  WGA_REC_PANIC = 6                /* the reference panics (errors.rs:92 slice, String::insert_str
                                      out of range at cigar.rs:507,513)                    */
};

#define WGA_NONE UINT64_MAX

/* Device-side per-record diagnostics (all fields WGA_NONE when clean).  Written with atomicMin so
 * "first in op order" / "first in reversed-sequence order" is exact. */
typedef struct {
  uint64_t bad_op_idx;   /* first op (index inside the record) the consumer rejects              */
  uint64_t panic_op_idx; /* first I/D op whose insertion point lies beyond the fetched sequence  */
  uint64_t bad_base_pos; /* first invalid base in reverse-complement order (utils.rs:85-98)      */
} wga_rec_diag;

/* = struct Cigar sans text (cigar.rs:16-29) */
typedef struct {
  uint64_t match, mismatch, ins_ev, ins_bp, del_ev, del_bp, inv_ins_ev, inv_ins_bp, inv_del_ev,
      inv_del_bp, inv_ev;
} wga_cigar_counts;

/* CSR batch of n records: d_ops[d_op_off[i] .. d_op_off[i+1]) belong to record i. */
typedef struct {
  const uint32_t* d_ops;
  const uint64_t* d_op_off; /* n+1 entries, d_op_off[0] == 0 */
  const uint8_t* d_strand_neg; /* n entries, 1 = '-' (common.rs:42-48) */
  uint64_t n_ops;              /* == d_op_off[n] (host copy, sizes the launch) */
  uint32_t n;
} wga_cigar_batch;

typedef struct wga_ctx wga_ctx;

/* ---- context / runtime plumbing ------------------------------------------------------------ */
int wga_abi_version(void);
const char* wga_last_error(void);
int wga_device_count(void);
/* THREADING / STREAM CONTRACT.  A context belongs to ONE host thread at a time and orders all its work on ONE
 * stream: its scratch arenas (scan partials, descriptors, coverage piece lists) are shared by every entry point and
 * reused from call to call, which is only safe because calls are stream-ordered.  Use one context per host thread /
 * per device; wga_ctx_set_stream / wga_ctx_reset_stream drain the stream they leave before switching.
 * wga_ctx_destroy waits for the context's stream. */
int wga_ctx_create(int device, wga_ctx** out);
void wga_ctx_destroy(wga_ctx*);
/* Launch on an external stream (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream;
 * NULL is HIP's default stream).  wga_ctx_reset_stream goes back to the context's own stream.  Both
 * synchronise the stream that is being left: an external stream must stay alive until the next
 * wga_ctx_set_stream / wga_ctx_reset_stream / wga_ctx_destroy on the context. */
int wga_ctx_set_stream(wga_ctx*, void* hip_stream);
int wga_ctx_reset_stream(wga_ctx*);
/* Tunables (test knobs): "expand_force_slow" (0/1) forces the u64 op-serial fallback of the
 * expand kernel; "expand_no_table" (0/1) forces its binary-search event lookup; "expand_variant" picks the row
 * kernel — same bytes either way: -1 (default) or 3: the streaming kernel, one wave per row kind of a run of tiles
 * (profiles/r04_k2s_experiments.md), 0: v1 (one block per tile; also the kernel of the tiles the streaming kernel leaves).  The
 * window kernel of rounds 3-5 (2) was retired in round 6: it was ahead only for batches below 100 ops per record
 * kernel (environment: WGA_EXPAND_VARIANT); "expand_job_tiles" (1 .. 32; 0, the default: 8, or 4 for batches below 200 000 tiles): consecutive tiles one wave of the
 * streaming kernel walks;
 * "pseudo_variant": wga_pafpseudo_fill's rows through the streaming kernel (3, default) or one block per tile (0);
 * "expand_drain_min" (0 .. 64, v1 only) = how many gap-touching 16-column chunks a wave
 * queues before it emits them: 0 (default) lets the library choose by the size of the two sequence pools — 32 when
 * they stay in the 256 MB Infinity Cache, 16 when they do not (environment: WGA_EXPAND_DRAIN_MIN).
 * "reduce_same_device_ok" / "reduce_staged" (0/1): wga_reduce_scatter_i32 over contexts that share a device, and by staged
 * peer copies where peer access exists (tests on one-GPU boxes).
 * "cov_spin_limit" (default 4096): wga_pafcov_accumulate's list pass polls a tile sum this often before it adds up the ops itself.
 * "op_long_ops" (default 16384) / "op_piece_ops" (8192, a multiple of 256): the op walks with one wave per record (call
 * events, chain lines, dotplot segments) cut records beyond the first into pieces of at most the second and walk the pieces
 * over the whole chip; "maf_long_cols" (32768; 61440 at most: longer blocks are long whatever it says) / "maf_piece_cols"
 * (16384): the same for the MAF column walks — a long block is listed, planned and walked in pieces on the device (no
 * read-back); when the configured piece size would make more than 32 768 pieces beyond one per long block, the piece grows
 * to whole steps of 2 048 columns that stay inside that budget.  "maf_group" (0 .. 8, default 0 = by the number of blocks:
 * n / 24 576, at least 1, at most 8): consecutive MAF blocks one wave walks as one column stream.  The tests set small values
 * to reach those paths with small inputs. */
int wga_ctx_set_param(wga_ctx*, const char* name, int64_t value);
/* Read back: "cov_spin_limit" (the setting), "expand_drain_min" = what the last wga_paf2maf_expand used,
 * "expand_variant" / "expand_job_tiles" / "pseudo_variant" (the settings), "expand_variant_used" (what the last
 * wga_paf2maf_expand ran), "expand_stream_left_to_v1" / "pseudo_stream_left_to_blocks" (tiles the streaming kernel's last
 * launch left to the block kernels: records whose slices do not match their CIGAR, slices at a pool's edge, giant tiles —
 * a device read, diagnostics; -1 once another call has taken the scratch arena the two counters lay in). */
int wga_ctx_get_param(wga_ctx*, const char* name, int64_t* value);
/* Measurement hook: after wga_ctx_set_param(ctx, "expand_timing", 1) every wga_paf2maf_expand
 * brackets its gap-insertion kernel (without the descriptor pre-pass) with two events on the
 * launch stream.  Returns the summed kernel time of the launches since the last call (at most the
 * 64 most recent) and their number; waits for them. */
int wga_ctx_expand_timing(wga_ctx*, double* ms_sum, uint32_t* launches);
int wga_sync(wga_ctx*);
int wga_malloc(wga_ctx*, size_t bytes, void** d_out);
int wga_free(wga_ctx*, void* d_ptr);
int wga_memcpy_h2d(wga_ctx*, void* d_dst, const void* h_src, size_t bytes);
int wga_memcpy_d2h(wga_ctx*, void* h_dst, const void* d_src, size_t bytes); /* synchronises */
int wga_memset(wga_ctx*, void* d_dst, int byte, size_t bytes);
/* Pinned host memory and an asynchronous device-to-host copy on the context's stream: what a writer needs to stream
 * a result (the MAF text of converter.rs:237-262, the BED lines of pafcov.rs:56-60) out of HBM in pieces while the
 * next piece is copied — wga_sync (or a later synchronising call) before the host reads h_dst. */
int wga_host_alloc(wga_ctx*, size_t bytes, void** h_out);
int wga_host_free(wga_ctx*, void* h_ptr);
int wga_memcpy_d2h_async(wga_ctx*, void* h_dst, const void* d_src, size_t bytes);

/* ---- host: CIGAR text -> packed ops (replaces the nom tokeniser, cigar.rs:43-75,
 *      utils.rs:69-74; driven per record by every consumer, e.g. cigar.rs:529-549) ------------
 * `text` is the CIGAR *after* the "cg:Z:" tag.  Emits ops until the first tokeniser error, which
 * is returned in *err (wga_rec_err) with the offending token in [*err_tok_off, +*err_tok_len).
 * Returns WGA_OK, or WGA_E_TOO_SMALL with *n_ops = required capacity. */
int wga_cigar_pack(const char* text, size_t len, uint32_t* ops, size_t cap, size_t* n_ops,
                   int32_t* err, size_t* err_tok_off, size_t* err_tok_len);
/* Upper bound of ops wga_cigar_pack can emit for `len` bytes of text. */
size_t wga_cigar_pack_bound(const char* text, size_t len);

/* ---- device tokeniser: the same text -> packed ops conversion as wga_cigar_pack, for callers that
 *      keep the CIGAR text on the device (the host tokeniser is the floor of
 *      every end-to-end run).  Record i's text (after the tag) is d_text[d_text_off[i] ..
 *      d_text_off[i+1]).  Two calls: with d_ops == NULL it fills d_op_cnt[n] (ops the record packs
 *      to, up to its first tokeniser error) and d_err[n]; after an exclusive scan of the counts the
 *      second call writes the ops at d_ops + d_op_off[i].  Error codes, token offsets and the
 *      splitting of lengths >= 2^28 are those of wga_cigar_pack. */
typedef struct {
  int32_t err;      /* wga_rec_err */
  uint32_t tok_len; /* offending token inside the record's text: length ... */
  uint64_t tok_off; /* ... and offset */
} wga_tok_err;
int wga_cigar_tokenise(wga_ctx*, uint32_t n, const uint8_t* d_text, const uint64_t* d_text_off,
                       uint64_t* d_op_cnt, wga_tok_err* d_err, uint32_t* d_ops,
                       const uint64_t* d_op_off);

/* ---- K1: PAF stat walk (replaces parse_paf_to_cigar, cigar.rs:629-707) ----------------------
 * d_counts[n]; d_diag[n] (bad_op_idx set for ops other than M = X I D).
 * d_tile_ws (optional, may be NULL): wga_tile_ws_bytes(n_ops) bytes, filled with the per-tile
 * partial sums wga_paf2maf_expand needs to place a tile inside a long record. */
size_t wga_tile_ws_bytes(uint64_t n_ops);
int wga_cigar_stat(wga_ctx*, const wga_cigar_batch*, wga_cigar_counts* d_counts,
                   wga_rec_diag* d_diag, void* d_tile_ws);

/* ---- paf2maf row geometry (host logic of converter.rs:196-263 moved on device) --------------
 * Row lengths follow String::insert_str semantics: t_row_len = t_src_len + I bases,
 * q_row_len = q_src_len + D bases (cigar.rs:504-515).  Each record occupies
 *   pre_t[i] | t row | pre_q[i] | q row | post[i]   bytes of the output text
 * (pre/post = the MAF line text around the rows, maf.rs:566-581; NULL = 0).
 * Outputs: d_t_row_off[n], d_q_row_off[n], d_rec_off[n+1] (d_rec_off[n] = total bytes). */
int wga_paf2maf_layout(wga_ctx*, uint32_t n, const wga_cigar_counts* d_counts,
                       const uint64_t* d_t_src_len, const uint64_t* d_q_src_len,
                       const uint32_t* d_pre_t, const uint32_t* d_pre_q, const uint32_t* d_post,
                       uint64_t* d_t_row_off, uint64_t* d_q_row_off, uint64_t* d_rec_off);

/* ---- K2: paf2maf gap insertion (replaces parse_cigar_to_insert + cigar_unit_insert_seq,
 *      cigar.rs:492-551, and reverse_complement, utils.rs:83-101) -----------------------------
 * t_fa/q_fa: ungapped forward-strand sequence pools; record i uses
 *   t_fa[t_src_off[i] .. +t_src_len[i])   and   q_fa[q_src_off[i] .. +q_src_len[i])
 * (what faidx fetch returned, converter.rs:219-225).  For strand '-' the kernel reads the query
 * slice reversed and complemented.  Writes the two gapped rows at d_out + d_*_row_off[i].
 * d_counts / d_tile_ws must come from wga_cigar_stat on the same batch; d_diag is updated
 * (panic_op_idx, bad_base_pos). */
int wga_paf2maf_expand(wga_ctx*, const wga_cigar_batch*, const wga_cigar_counts* d_counts,
                       const void* d_tile_ws, const uint8_t* d_t_fa, uint64_t t_fa_bytes,
                       const uint64_t* d_t_src_off, const uint64_t* d_t_src_len,
                       const uint8_t* d_q_fa, uint64_t q_fa_bytes, const uint64_t* d_q_src_off,
                       const uint64_t* d_q_src_len, uint8_t* d_out, const uint64_t* d_t_row_off,
                       const uint64_t* d_q_row_off, wga_rec_diag* d_diag);

/* Copy n variable-length byte snippets: d_dst[d_dst_off[i] ..) = d_src[d_src_off[i] .. d_src_off[i+1])
 * (used to drop the "a score=…" / "s\tname\t…" line text between the rows, maf.rs:566-581). */
int wga_scatter_bytes(wga_ctx*, uint32_t n, const uint8_t* d_src, const uint64_t* d_src_off,
                      uint8_t* d_dst, const uint64_t* d_dst_off);

/* ---- K3: MAF column-pair walk (replaces parse_maf_seq_to_cigar + cigar_cat_ext,
 *      cigar.rs:298-308,344-432) ---------------------------------------------------------------
 * Record i compares d_rows[t_off[i] + j] with d_rows[q_off[i] + j] for j < cols[i]
 * (cols = min of the two row lengths: `zip` truncates).  d_counts[n] as K1.
 * Optional run list for maf2paf's cg:Z: text (maf.rs:484-520): d_run_cnt[n] always receives the
 * number of runs of record i; if d_runs != NULL record i's runs are written in order to
 * d_runs[d_run_off[i] ..) as (start_column << 3 | class), class 0 '=', 1 'I', 2 'D', 3 'X'
 * (a run's length is the next run's start — or cols[i] — minus its own start).  Call once with
 * d_runs == NULL to size, scan d_run_cnt (wga_exclusive_scan_u64), call again. */
int wga_maf_pair_stat(wga_ctx*, uint32_t n, const uint8_t* d_rows, const uint64_t* d_t_off,
                      const uint64_t* d_q_off, const uint64_t* d_cols,
                      const uint8_t* d_strand_neg, wga_cigar_counts* d_counts,
                      uint64_t* d_run_cnt, uint64_t* d_runs, const uint64_t* d_run_off);

/* ---- K4: MAF call column walk (replaces the group_by(cigar_cat_ext_caller) scan of
 *      call_within_var, caller.rs:444-446, and the per-column gap scans of
 *      find_safe_chunk_boundary :173-199 / create_chunk_record :240-251) -------------------------
 * Ordered list of maximal runs of equal caller class per record (cigar.rs:314-328), 3 u64 per
 * run: [0] start_column << 3 | class (0 '=', 1 I, 2 D, 3 X, 4 W = both rows gapped),
 * [1] non-gap target characters before the run, [2] non-gap query characters before it.
 * Same two-call protocol as wga_maf_pair_stat (d_runs == NULL sizes; d_run_off counts runs). */
int wga_maf_call_runs(wga_ctx*, uint32_t n, const uint8_t* d_rows, const uint64_t* d_t_off,
                      const uint64_t* d_q_off, const uint64_t* d_cols, uint64_t* d_run_cnt,
                      uint64_t* d_runs, const uint64_t* d_run_off);

/* ---- call on PAF: the op walk (replaces the fold of call_within_var_paf, caller.rs:664-819) ---
 * Walks each record's ops with the running target / query positions and the `after_m` flag of
 * the reference and lists the ops that raise VCF rows: every X op when `snp` is set (one row per
 * column, :688-717), and every I / D op that directly follows an M / = / X op and is longer than
 * `svlen` (:719-813).  The walk of a record stops at its first op outside M = X I D (the
 * reference's fold keeps the CigarOpInvalid in its accumulator, skips the remaining ops and then
 * discards the error, :673,815-819).  An I / D whose length was split by the packer (>= 2^28) is
 * listed when a continuation piece follows; the caller applies the cutoff to the summed length.
 * Event = 3 u64: op index within the record, target bases before the op, query bases before it
 * (both relative to the record's start coordinates).  Same two-call protocol as wga_maf_call_runs:
 * d_ev == NULL -> only d_ev_cnt[n]; then d_ev_off = exclusive scan and the call again. */
int wga_paf_call_events(wga_ctx*, const wga_cigar_batch*, uint64_t svlen, int snp, uint64_t* d_ev_cnt,
                        uint64_t* d_ev, const uint64_t* d_ev_off);

/* ---- call on PAF: the VCF rows of the events (replaces the record building and printing of call_within_var_paf,
 *      caller.rs:640-658 the <INV> row of a '-' record, :688-717 one row per column of an X op, :719-813 the INS / DEL
 *      rows of indels longer than `svlen`; text layout of noodles-vcf 0.43, README.md:323-343) ---------------------
 * Record i's text is its rows in the reference's order:
 *   <target>\t<pos>\t.\t<REF>\t<ALT>\t.\t.\t<INFO or .>\tGT:QI\t1|1:<query>@<a>[@<b>]@<P|N>\n
 * d_ev / d_ev_off: the events of wga_paf_call_events on the same batch (with the same svlen).  d_recs: per record its
 * names (offsets into d_names), PAF coordinates and the place of its fetched target / query sequence inside the pools
 * (paf.rs:221-237: [start, end] inclusive, clipped at the contig end).  Two calls: d_out == NULL -> d_nbytes[n] and
 * d_err[n]; then d_out_off = exclusive scan of d_nbytes and the call again with d_out.  d_err[i].item == ~0: clean;
 * otherwise the record's first failing item (0 = the <INV> row, 1 + e = event e) with kind 1 = a REF / ALT slice outside
 * the fetched sequence (the reference's slice panic, caller.rs:695-696,753-754,800-801) or kind 2 = a base outside
 * ACGTN in any case (noodles-vcf's parse error; ch = the byte); the record's text ends in front of that item. */
typedef struct {
  uint64_t t_name_off, q_name_off;
  uint32_t t_name_len, q_name_len;
  uint64_t t_start, t_end, q_start, q_end;
  uint64_t t_off, t_len, q_off, q_len;
} wga_vcf_rec;
typedef struct {
  uint64_t item;
  uint32_t kind, ch;
} wga_vcf_err;
int wga_paf_call_vcf(wga_ctx*, const wga_cigar_batch*, uint64_t svlen, const uint64_t* d_ev, const uint64_t* d_ev_off,
                     const wga_vcf_rec* d_recs, const uint8_t* d_names, const uint8_t* d_t_pool,
                     const uint8_t* d_q_pool, uint64_t* d_nbytes, wga_vcf_err* d_err, uint8_t* d_out,
                     const uint64_t* d_out_off);

/* ---- call on MAF: chunk cuts, event rules and VCF rows on the run list (replaces find_safe_chunk_boundary
 *      caller.rs:159-219, create_chunk_record :221-265 and the fold + record building of call_within_var :388-608;
 *      text layout of noodles-vcf 0.43, README.md:323-343) ---------------------------------------------------------
 * d_rows / d_t_off / d_q_off / d_cols / d_runs / d_run_off: the arrays of wga_maf_call_runs (cols = the target row's
 * length, caller.rs:115) and the run list it wrote.  d_recs: per block its names (offsets into d_names), the two s lines'
 * start fields, the query's source size and strand (maf.rs:65-73).  Block i's text = the rows of its chunks in order: the
 * <INV> row of a '-' block's chunk (`inv`, :423-440), one row per column of an X run (`snp`, :570-603), one row per I / D
 * run longer than `svlen` that follows an '=' / X run (:464-569).  Two calls as wga_paf_call_vcf: d_out == NULL ->
 * d_nbytes[n], d_err[n] (item == ~0: clean; kind 2 = a REF / ALT base outside ACGTN in any case, ch = the byte:
 * noodles-vcf's parse error); then d_out_off = exclusive scan of d_nbytes and the call again with d_out.  A block's text
 * ends in front of the chunk that holds its first bad base (a chunk's records are collected before any is written, :137-141). */
typedef struct {
  uint64_t t_name_off, q_name_off;
  uint32_t t_name_len, q_name_len;
  uint64_t t_start, q_start, q_size;
  uint32_t q_neg, pad;
} wga_maf_vcf_rec;
int wga_maf_call_vcf(wga_ctx*, uint32_t n, const uint8_t* d_rows, const uint64_t* d_t_off, const uint64_t* d_q_off,
                     const uint64_t* d_cols, const uint64_t* d_runs, const uint64_t* d_run_off, const wga_maf_vcf_rec* d_recs,
                     const uint8_t* d_names, int snp, int inv, uint64_t svlen, uint64_t chunk_size, uint64_t* d_nbytes,
                     wga_vcf_err* d_err, uint8_t* d_out, const uint64_t* d_out_off);

/* ---- BGZF inflate on the device (the reference opens bgzipped FASTA through htslib's faidx,
 *      converter.rs:183-184, paf.rs:221-237, pseudomaf.rs:214-237) -------------------------------------------------
 * d_in: the compressed file as it is; d_blocks[n]: per BGZF member the place of its raw DEFLATE stream (behind the gzip
 * header with the BC field, in front of the CRC32 / ISIZE trailer), ISIZE and the place of its bytes in the output — the
 * host walks the member headers, nothing else.  Every block is inflated by one wave (RFC 1951: stored, fixed and dynamic
 * blocks) into d_out + out_off.  d_status[n]: 0, or the reason a stream is corrupt (WGA_INF_* in the kernel source: the
 * input ends inside a symbol, a bad block type / stored length, bad code lengths, an unused bit pattern, a distance in front
 * of the block, a size other than ISIZE); a corrupt block never writes outside its own out_len bytes.  The CRC32 of a
 * member is not checked (neither does the host reader this replaces). */
typedef struct {
  uint64_t in_off;
  uint32_t in_len, out_len;
  uint64_t out_off;
} wga_bgzf_block;
int wga_bgzf_inflate(wga_ctx*, const uint8_t* d_in, uint64_t in_bytes, uint32_t n_blocks, const wga_bgzf_block* d_blocks,
                     uint8_t* d_out, uint32_t* d_status);

/* ---- paf2chain: the data lines of parse_cigar_to_chain + cigar_unit_chain
 *      (cigar.rs:251-295,460-490) and the head / tail indel trim of parse_cigar_to_trim
 *      (cigar.rs:202-245) that the chain header needs (chain.rs:142-183) ----------------------------
 * Record i's text is "\n<size>\t<dt>\t<dq>" per block but the last, then "\n<size>"; the caller puts
 * the header in front and "\n\n" behind (converter.rs:148-173).  Two calls: with d_out == NULL the
 * kernel fills d_trim[n], d_nbytes[n] (text bytes of record i) and d_diag[n].bad_op_idx (first op
 * outside M = X I D: CigarOpInvalid); the second call writes record i's text at d_out + d_out_off[i]. */
typedef struct {
  uint64_t head_ins, head_del, tail_ins, tail_del;
} wga_chain_trim_t;
int wga_cigar_chain(wga_ctx*, const wga_cigar_batch*, wga_chain_trim_t* d_trim, uint64_t* d_nbytes,
                    wga_rec_diag* d_diag, uint8_t* d_out, const uint64_t* d_out_off);

/* ---- K11: bridges between run / data-line lists, packed ops and CIGAR text ----
 * The remaining chain converters and maf2paf's text all reduce to walks that already exist once
 * their input is a wga_cigar_batch, and their CIGAR text is pure integer formatting:
 *   maf2chain  (converter.rs:57-91, parse_maf_seq_to_chain / _trim cigar.rs:155-199,435-457)
 *              = K3 runs -> wga_maf_runs_ops -> wga_cigar_chain  (cigar_cat merges '=' and X into M:
 *              cigar_unit_chain adds adjacent M-like ops into one size, cigar.rs:467-476)
 *   maf2paf    cg:Z: text (maf.rs:484-520, cigar.rs:400-401)          = wga_maf_runs_cigar_text
 *   chain2maf  (converter.rs:268-358, parse_chain_to_insert :360-388) = wga_chain_lines_ops -> K1 -> K2
 *   chain2paf  (chain.rs:430-452, parse_chain_to_cigar cigar.rs:554-627)
 *              = wga_chain_lines_ops -> K1 (matches, block length) + wga_chain_lines_cigar_text
 * A data line is three u64: size, 2nd text column (D bases), 3rd text column (I bases); a line is
 * walked as M size, I, D (zero lengths are left out of the ops; the text always has "<size>M").
 * Runs are K3's (wga_maf_pair_stat).  n_elems = d_*_off[n] = total runs / lines (< 2^32).
 * Two calls each: d_out == NULL fills d_cnt[n] (ops / bytes of record i); then record i's output is
 * written at d_out + d_out_off[i] (ops: in elements; text: in bytes).  The count call's scan of the
 * element sizes stays in the context for the fill call with the same arrays and counts (8 bytes per
 * element, grow-only).  It is one-shot (the fill call that takes it consumes it; a second fill call computes its own) and
 * is dropped when wga_free, wga_memcpy_h2d or wga_memset touches one of the arrays.  Contents changed behind the library's
 * back (another stream, another library) between the two calls are the caller's to fence with a new count call. */
int wga_maf_runs_ops(wga_ctx*, uint32_t n, uint64_t n_elems, const uint64_t* d_runs, const uint64_t* d_run_off,
                     const uint64_t* d_cols, uint64_t* d_cnt, uint32_t* d_out, const uint64_t* d_out_off);
int wga_maf_runs_cigar_text(wga_ctx*, uint32_t n, uint64_t n_elems, const uint64_t* d_runs,
                            const uint64_t* d_run_off, const uint64_t* d_cols, uint64_t* d_cnt, uint8_t* d_out,
                            const uint64_t* d_out_off);
int wga_chain_lines_ops(wga_ctx*, uint32_t n, uint64_t n_elems, const uint64_t* d_lines,
                        const uint64_t* d_line_off, uint64_t* d_cnt, uint32_t* d_out, const uint64_t* d_out_off);
int wga_chain_lines_cigar_text(wga_ctx*, uint32_t n, uint64_t n_elems, const uint64_t* d_lines,
                               const uint64_t* d_line_off, uint64_t* d_cnt, uint8_t* d_out,
                               const uint64_t* d_out_off);

/* ---- K12: dotplot base-level segments (replaces the fold over
 *      emit_baseplotdatas, cigar.rs:815-914, of parse_cigar_to_base_plotdata :917-952 and — on ops
 *      from wga_maf_runs_ops — parse_maf_to_base_plotdata :955-985) ----------------------------------
 * Per record the ordered BasePlotdata list (dotplot.rs:181-190) without the two names: 5 u64 per
 * segment = ref_start, ref_end, query_start, query_end (start / end swapped for '-' records as
 * reserve_query_start_end does), kind 0 'M' / 1 'I' / 2 'D'.  An I / D longer than `cutoff` is its
 * own segment; M-like ops and shorter indels merge into M segments; other ops are ignored.
 * d_t_start / d_q_start = target_start() / query_start() of the records.  Two calls: d_segs == NULL
 * fills d_seg_cnt[n]; then record i's segments go to d_segs + 5 * d_seg_off[i]. */
int wga_cigar_dotplot(wga_ctx*, const wga_cigar_batch*, uint64_t cutoff, const uint64_t* d_t_start,
                      const uint64_t* d_q_start, uint64_t* d_seg_cnt, uint64_t* d_segs,
                      const uint64_t* d_seg_off);

/* ---- K13: PAF field splitter (replaces the csv / serde record reader of
 *      paf.rs:24-30,50-78 and the tag lookup of get_cigar_string :122-141 for plain files) ---------
 * d_text = the PAF file as read (n_bytes < 4 GiB).  One wga_paf_line per text line, in order:
 * status WGA_PAF_OK (a record: the 12 fixed fields parsed like u64::from_str / Strand::from_str,
 * name spans and the span of the cg:Z: text as byte offsets into d_text — cg_beg == WGA_NONE when the
 * record has no cg:Z: tag), WGA_PAF_SKIP (blank line or '#' comment, which the csv reader drops) or
 * WGA_PAF_FALLBACK (a '"' or CR on the line, fewer than 12 fields, a field that does not parse, or
 * a cs:Z: tag standing in for cg:Z:): if any line says FALLBACK the caller must read the file with a
 * csv-semantics parser, which yields the same records or the reference's error text.
 * Two calls: d_lines == NULL returns *n_lines (host value; the call synchronises); then
 * d_lines[cap_lines >= *n_lines] is filled.  The cg spans feed wga_cigar_tokenise_spans, the
 * tokeniser on texts that are not back to back (record i = d_text[d_beg[i], d_end[i])). */
#define WGA_PAF_OK 0
#define WGA_PAF_SKIP 1
#define WGA_PAF_FALLBACK 2
typedef struct {
  uint64_t num[9]; /* query_length, query_start, query_end, target_length, target_start, target_end,
                      matches, block_length, mapq (paf.rs:50-65) */
  uint64_t qname_off, tname_off, cg_beg, cg_end;
  uint32_t qname_len, tname_len, n_fields;
  uint8_t strand_neg, status, pad[2];
} wga_paf_line;
int wga_paf_split(wga_ctx*, const uint8_t* d_text, uint64_t n_bytes, uint64_t* n_lines, wga_paf_line* d_lines,
                  uint64_t cap_lines);
int wga_cigar_tokenise_spans(wga_ctx*, uint32_t n, const uint8_t* d_text, const uint64_t* d_beg,
                             const uint64_t* d_end, uint64_t* d_op_cnt, wga_tok_err* d_err, uint32_t* d_ops,
                             const uint64_t* d_op_off);

/* ---- K14: MAF line splitter (replaces MAFReader / parse_sline for plain files, maf.rs:25-36,138-211,
 *      371-421; with it the K3 / K4 walks read the rows straight out of the uploaded file) -----------
 * One wga_maf_line per text line: WGA_MAF_SLINE (a line starting with 's' other than the file's first
 * line, split at white space into mode, name, start, size, strand, srcSize, text: numbers parsed like
 * u64::from_str / Strand::from_str, name and text as byte spans of d_text), WGA_MAF_OTHER (any other
 * line: it ends the block in progress) or WGA_MAF_FALLBACK (an s-line without exactly seven tokens,
 * with a field that does not parse or with a non-ASCII byte): the caller then reads the file with its
 * host reader, which returns the reference's error.  A block is a maximal run of s-lines.  Same
 * two-call protocol and size limit as wga_paf_split. */
#define WGA_MAF_SLINE 0
#define WGA_MAF_OTHER 1
#define WGA_MAF_FALLBACK 2
typedef struct {
  uint64_t num[3]; /* start, align_size, size (maf.rs:65-73) */
  uint64_t name_off, seq_off, seq_len;
  uint32_t name_len;
  uint8_t strand_neg, status, pad[2];
} wga_maf_line;
int wga_maf_split(wga_ctx*, const uint8_t* d_text, uint64_t n_bytes, uint64_t* n_lines, wga_maf_line* d_lines,
                  uint64_t cap_lines);

/* ---- stat totals: d_totals[11] = the column sums of d_counts[n] (wga_cigar_counts order).  What
 *      Statistic::merge adds up for one (ref, query) pair (stat.rs:181-223); with records sharded over
 *      GPUs these 88 bytes are the only thing `stat` has to all-reduce.  d_totals is overwritten. */
int wga_counts_total(wga_ctx*, uint32_t n, const wga_cigar_counts* d_counts, uint64_t* d_totals);

/* ---- FASTA text in HBM -> sequence pool (replaces the htslib faidx fetches of converter.rs:183-184,219-225,
 *      paf.rs:221-237, pseudomaf.rs:214-237: the drivers hand (contig, start, length) as an offset into the pool) ----
 * d_text = the FASTA file as read (decompressed if it was bgzf / gzip).  A header line starts with '>' at a line start;
 * the pool holds every byte of the sequence lines behind the first header, minus '\n' and a '\r' in front of one, case
 * preserved, contigs back to back in file order.  Two calls:
 *   d_pool == NULL: counts -> *n_contigs, *pool_bytes (host values; synchronises).
 *   otherwise     : fills d_pool[pool_bytes] and d_contigs[n_contigs]; a contig's name is the text behind '>' at
 *                   hdr_start up to the first white space (the host reads it from its own copy of the text).
 * Lookups keep htslib's semantics on the host side: first of duplicate names, end inclusive, clipped to the contig. */
typedef struct {
  uint64_t hdr_start; /* offset of the '>'                                          */
  uint64_t hdr_end;   /* offset of the header line's '\n' (n_bytes if the file ends) */
  uint64_t pool_off;  /* first base of the contig in the pool                        */
  uint64_t len;       /* bases                                                       */
} wga_fa_contig;
int wga_fasta_pool(wga_ctx*, const uint8_t* d_text, uint64_t n_bytes, uint64_t* n_contigs, uint64_t* pool_bytes,
                   uint8_t* d_pool, wga_fa_contig* d_contigs);

/* ---- K5: pafcov (replaces update_cov_vec, cigar.rs:710-741, and the per-thread array merge of
 *      pafcov.rs:29-53) ------------------------------------------------------------------------
 * Record i adds +1 to d_cov[cov_off[target_id[i]] + p] for every base p of its M / = ops that
 * lies below cov_len[target_id[i]].  Implemented as a difference array: accumulate() adds the
 * ±1 marks, finalize() turns marks into counts by an inclusive scan per target.  `total_cov` =
 * number of counters in d_cov (max over targets of cov_off + cov_len).  accumulate() may be
 * called for several batches before finalize(); it waits for its list pass twice (it reads two
 * counts back to size its work lists).  The context keeps its work lists between calls, grow-only:
 * 256 bytes per 1024 ops of the largest batch (the pieces as the list pass writes them) and 40 bytes
 * per piece (a record segment of a tile of ops under a window of 8192 counters; 3.5 per 1024 ops on
 * configs[3]'s records).  Context parameter "cov_spin_limit": polls of a neighbouring tile's sum
 * before a wave of the list pass adds up the ops itself (default 4096; 0: always adds them up).
 *
 * accumulate_final() = accumulate() of the LAST batch + finalize() in one pass over the array: the
 * kernel that replays the batch's marks window by window goes on to scan each window and hands its
 * sum to the windows behind it (decoupled look-back), so the array is read and written once instead
 * of twice more.  The targets' ranges [cov_off[t], cov_off[t] + cov_len[t]) must not overlap (any
 * order; WGA_E_INVALID_ARG otherwise — also from finalize()); counters between them are not touched.
 * The merge across the reference's per-thread arrays (pafcov.rs:29-53) has no counterpart: there is
 * one array. */
int wga_pafcov_accumulate(wga_ctx*, const wga_cigar_batch*, const uint32_t* d_target_id,
                          const uint64_t* d_t_start, const uint64_t* d_cov_off,
                          const uint64_t* d_cov_len, int32_t* d_cov, uint64_t total_cov);
int wga_pafcov_finalize(wga_ctx*, uint32_t n_targets, const uint64_t* d_cov_off,
                        const uint64_t* d_cov_len, int32_t* d_cov);
int wga_pafcov_accumulate_final(wga_ctx*, const wga_cigar_batch*, const uint32_t* d_target_id,
                                const uint64_t* d_t_start, const uint64_t* d_cov_off,
                                const uint64_t* d_cov_len, uint32_t n_targets, int32_t* d_cov,
                                uint64_t total_cov);

/* K18 — output bytes in HBM -> BGZF: what `-o out.maf.gz` stands for in the reference (utils.rs:181-228 wraps the output
 * file in flate2's GzEncoder, level 6, when the name ends in `.gz`; converter.rs / pafcov.rs / pseudomaf.rs write through
 * that writer).  The promise to a reader is a gzip stream that inflates to the plain output; this entry keeps it with the
 * text still on the device, so the compressed bytes are what crosses PCIe.  The stream is a sequence of complete gzip
 * members of at most 32 768 input bytes, each with the BGZF `BC` extra field (members are independent: seekable by
 * htslib-style readers, concatenable; any gzip reader takes the whole), each one deflate block of literals under the
 * member's own Huffman code, or a stored block where that is not smaller.  Calls on consecutive pieces of an output
 * concatenate into one valid stream; eof_marker != 0 appends BGZF's 28-byte empty member behind the last piece.
 *   wga_bgzf_bound(n)      what d_out must hold for n input bytes in the worst case (stored members + marker)
 *   wga_bgzf_compress(..)  d_in[n_bytes] -> d_out[*out_bytes]; synchronises once (the size); d_in, d_out any alignment.
 *                          d_out == NULL: the count call — *out_bytes is the exact size of the stream, nothing is written
 *                          (a caller that cannot spare wga_bgzf_bound's worst case allocates by it and calls again).
 *                          WGA_E_INVALID_ARG with *out_bytes set when out_cap is too small (nothing written). */
uint64_t wga_bgzf_bound(uint64_t n_bytes);
int wga_bgzf_compress(wga_ctx*, const uint8_t* d_in, uint64_t n_bytes, uint8_t* d_out, uint64_t out_cap, uint64_t* out_bytes,
                      int eof_marker);

/* pafcov's text back end (pafcov.rs:56-60): the BED lines
 * "<name>\t<pos>\t<pos+1>\t<count>\n" for positions p0 .. p0+count-1 of one target, d_cov pointing
 * at the counter of p0.  Two calls: with d_out == NULL it fills d_line_off[count+1] (exclusive
 * scan of the line lengths, total bytes in the last entry); the second call writes line k at
 * d_out + d_line_off[k]. */
int wga_pafcov_format(wga_ctx*, const uint8_t* d_name, uint32_t name_len, const int32_t* d_cov,
                      uint64_t p0, uint32_t count, uint64_t* d_line_off, uint8_t* d_out);

/* ---- per-record class sums of a batch: bases in M/=/X, I, D, S and "other" (N H P ...) ops.
 *      pafpseudo's host logic (pseudomaf.rs:147-202) needs M+X+D (target span) and the edited
 *      query length q_len - (I+S) + D before it can place segments.  d_sums: n x 5 u64.
 *      The call is the count call of pafpseudo's protocol: its tile and record sums stay in the
 *      context (88 bytes per 1024 ops + 40 per record, grow-only) for wga_pafpseudo_fill on the
 *      same batch arrays, which then does not compute them again (one-shot: that fill call consumes
 *      them; dropped when the op arrays are written through wga_memcpy_h2d / wga_memset or freed). --- */
typedef struct {
  uint64_t mx, i, d, s, o;
} wga_class_sums;
int wga_cigar_class_sums(wga_ctx*, const wga_cigar_batch*, wga_class_sums* d_sums);

/* ---- K6: pafpseudo row segments (replaces gen_pesudo_maf_by_cigar, cigar.rs:744-804, plus the
 *      overlap trim of pseudomaf.rs:190-192) ---------------------------------------------------
 * Record i writes its target-coordinate segment (length M+X+D, minus skip[i] leading columns)
 * at d_out + d_dst_off[i].  base_mode = 1: bases from q_fa (reverse-complemented for '-'),
 * 0: symbols '1' (M/=), '0' (X), '-' (D). */
int wga_pafpseudo_fill(wga_ctx*, const wga_cigar_batch*, int base_mode, const uint8_t* d_q_fa,
                       uint64_t q_fa_bytes, const uint64_t* d_q_src_off,
                       const uint64_t* d_q_src_len, const uint64_t* d_skip, uint8_t* d_out,
                       const uint64_t* d_dst_off, wga_rec_diag* d_diag);

/* ---- several devices in one process: the coverage reduce of a target whose records are spread over the devices
 *      (the element-wise merge of the reference's per-thread arrays, pafcov.rs:29-53, moved onto xGMI) ---------------------
 * ctxs[g] = a context on device g (ngpu distinct devices), d_bufs[g] = `count` int32 counters on that device.  After the
 * call, slice g of the index space — [count * g / ngpu, count * (g + 1) / ngpu) — of d_bufs[g] holds the element-wise sum
 * over all devices (a reduce-scatter); the rest of every buffer is unchanged.  Every device pulls its slice from every
 * other device (hipMemcpyPeer: all links busy at once, no ring) and adds it; the call synchronises all the contexts
 * before and after.  One host thread drives it (the contexts' owner threads must be idle). */
int wga_reduce_scatter_i32(wga_ctx** ctxs, int ngpu, int32_t** d_bufs, uint64_t count);

/* ---- utility: exclusive scan of n u64 values on device (d_out[n+1], d_out[n] = total) ------- */
int wga_exclusive_scan_u64(wga_ctx*, uint32_t n, const uint64_t* d_in, uint64_t* d_out);

#ifdef __cplusplus
}
#endif
#endif /* WGA_HIP_H */
