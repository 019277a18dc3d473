/*
 * wga_capi.cpp — the C-ABI of libwgahip.so (include/wga_hip.h): context, memory plumbing and
 * the launch logic of every kernel.  Compiled as HIP for gfx950 (product) or, with -DWGA_EMU,
 * as plain C++ over tests/emu/simt_emu.h (CPU logic tests only).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

/* the kernels, one header per family (each names the reference code it replaces), in dependency order: a header may use helpers
 * of the ones in front of it */
#include <type_traits>

#include "wga_kernels.h"       /* K1 stat, v1 of the row kernel, scans, layout */
#include "wga_kernels_k2s.h"   /* K2s: the streaming row kernel (paf2maf, pafpseudo's rows) */
#include "wga_k_class.h"
#include "wga_k5_pafcov.h"
#include "wga_k6_pafpseudo.h"
#include "wga_k3_maf.h"        /* K3 / K4: the MAF walks */
#include "wga_k7_paf_call.h"
#include "wga_k8_tokenise.h"
#include "wga_k9_bed.h"
#include "wga_k10_chain.h"
#include "wga_k11_bridges.h"
#include "wga_k12_dotplot.h"
#include "wga_k13_splitters.h"
#include "wga_k15_fasta.h"
#include "wga_k18_bgzf_deflate.h"
#include "wga_kernels3.h"      /* K16 VCF rows of call on PAF, K17 BGZF inflate */
#include "wga_k19_maf_call.h"  /* K19: rules and VCF rows of call on MAF */

struct wga_ctx {
  int device = 0;
  wga_stream_t own_stream = nullptr;
  wga_stream_t stream = nullptr;
  int expand_force_slow = 0;
  int expand_no_table = 0;
  unsigned expand_drain_min = 0; /* v1 only: 0 = by the size of the pools (WGA_DRAIN_POOL_BYTES: 32, 16 for genome-sized pools) */
  unsigned expand_drain_min_used = 0;
  uint64_t op_long_ops = 16384;  /* op walks with one wave per record (K7 call events, K12 dotplot segments): records beyond this many ops ... */
  uint64_t op_piece_ops = 8192;  /* ... are walked in pieces of this many (a multiple of 256), one wave each (test knobs: "op_long_ops", "op_piece_ops") */
  uint64_t maf_long_cols = 32768;  /* MAF blocks beyond this many columns are walked piece by piece ... */
  uint64_t maf_piece_cols = 16384; /* ... of this many columns, one wave each (test knobs: "maf_long_cols", "maf_piece_cols") */
  unsigned maf_group = 0;          /* blocks per wave of the MAF stream kernels, 1 .. 8 ("maf_group"; 0 = by the number of blocks) */
  void* maf_tab = nullptr;         /* K3 / K4: the table of a call's long blocks (header, list, pieces), grow-only */
  size_t maf_tab_cap = 0;
  bool maf_hdr_clean = false;      /* the header's append counters are zero (the plan kernel leaves them so) */
  struct MafKey { /* what the table was built from: a count call leaves it for the fill call on the same arrays (one shot) */
    bool valid = false, caller = false;
    uint32_t n = 0;
    const void *rows = nullptr, *t_off = nullptr, *q_off = nullptr, *cols = nullptr;
    uint64_t long_cols = 0, piece_cols = 0;
    bool same(const MafKey& o) const {
      return caller == o.caller && n == o.n && rows == o.rows && t_off == o.t_off && q_off == o.q_off && cols == o.cols &&
             long_cols == o.long_cols && piece_cols == o.piece_cols;
    }
  } maf_key;
  int expand_variant = -1; /* the row kernel: -1 / 3 the streaming kernel (wga_kernels_k2s.h), 0 v1 (wga_kernels.h: what the
                              streaming kernel leaves is v1's in either case).  The window kernel of rounds 3-5 (2) is gone: it was
                              ahead only below 100 ops per record (30-op records 6.1 against 7.3 ms) */
  int expand_variant_used = 0;
  int expand_job_tiles = 0; /* streaming kernel: tiles per wave ("expand_job_tiles"); 0 = by the batch (job_tiles_for) */
  int pseudo_variant = 3;   /* pafpseudo's rows: 3 the streaming row kernel, 0 one block per tile ("pseudo_variant") */
  const u32* pseudo_counts = nullptr; /* ... the two counters of the tiles its last launch left to the block kernel */
  const u32* stream_counts = nullptr; /* streaming kernel: the two counters of the tiles its last launch left to v1 (in the scratch arena) */
  void* scratch = nullptr;
  size_t scratch_cap = 0;
  /* the piece table of the op walks over long records (K7, K10, K12): built by the count call of the two-call protocol and
   * kept for the fill call on the same batch with the same parameters (the key), in a buffer of its own (grow-only) */
  struct OpTabKey {
    int kernel = 0;
    const void *ops = nullptr, *op_off = nullptr, *x0 = nullptr, *x1 = nullptr, *x2 = nullptr;
    uint32_t n = 0;
    uint64_t n_ops = 0, p0 = 0, p1 = 0, long_ops = 0, piece_ops = 0;
    bool operator==(const OpTabKey& o) const {
      return kernel == o.kernel && ops == o.ops && op_off == o.op_off && x0 == o.x0 && x1 == o.x1 && x2 == o.x2 && n == o.n &&
             n_ops == o.n_ops && p0 == o.p0 && p1 == o.p1 && long_ops == o.long_ops && piece_ops == o.piece_ops;
    }
  };
  struct OpTab {
    void* mem = nullptr;
    size_t cap = 0;
    OpTabKey key;
    bool valid = false;
    uint32_t np = 0;
    bool all = false;         /* every record is in the table (the one-wave kernel is not launched) */
    u64* piece_off = nullptr; /* n + 1 */
    u32* piece_rec = nullptr; /* np: the record of every piece */
    void* pieces = nullptr;   /* np x per_piece bytes */
  } op_tab;
  struct ClassTab { /* pafpseudo: the tile and record class sums of wga_cigar_class_sums, kept for wga_pafpseudo_fill */
    void* mem = nullptr;
    size_t cap = 0;
    bool valid = false;
    const void *ops = nullptr, *op_off = nullptr;
    uint32_t n = 0;
    uint64_t n_ops = 0;
    wga_tile_sum* tiles = nullptr;
    wga_class_sums* rec_sums = nullptr;
  } class_tab;
  struct ElemScan { /* K11: the count call's scan of the element sizes, kept for the fill call (grow-only buffer) */
    void* mem = nullptr;
    size_t cap = 0;
    bool valid = false;
    int kind = 0;
    const void* elem_off = nullptr;
    uint32_t n = 0, ne = 0;
    unsigned char src[32] = {0}; /* the entry point's source arrays (its functor) */
  } elem_scan;
  void* cov_pieces = nullptr; /* pafcov: the pieces' descriptors (wga_cov_desc) in window order, grow-only */
  u64 cov_pieces_cap = 0;
  void* cov_tile_list = nullptr; /* pafcov: WGA_COV_TILE_CAP piece slots per tile of ops, grow-only */
  u64 cov_tile_list_cap = 0;     /* in tiles */
  void* cov_order = nullptr;  /* pafcov: the order the marks -> counts replay takes the windows in, kept for the ranges it was made for */
  u64 cov_order_cap = 0;
  std::vector<u64> cov_order_key;
  void* cov_list = nullptr;   /* pafcov: the pieces beyond a tile's slots (WGA_COV_LISTS regions of cov_list_rcap) */
  u64 cov_list_rcap = 0;
  /* wga_reduce_scatter_i32: events that order this context's stream against the other devices' (created at first use), a
   * stream per staged pull, and what the two test switches say */
  bool rs_have_ev = false;
  rt_event_t rs_ready, rs_done;
  std::vector<wga_stream_t> rs_streams;
  std::vector<rt_event_t> rs_copied;
  bool rs_same_device_ok = false; /* "reduce_same_device_ok": distinct contexts may share a device (one-GPU test boxes) */
  bool rs_staged = false;         /* "reduce_staged": pull into scratch over N-1 streams instead of reading the peers in place */
#ifdef WGA_EMU
  u32 cov_spin_limit = 64; /* the emulator runs one block at a time: a tile that is not there yet will not come while this one polls */
#else
  u32 cov_spin_limit = 1u << 12;
#endif /* polls of a tile sum (milliseconds of waiting where ten microseconds are the rule) before the
                                    look-back adds up the ops itself (parameter "cov_spin_limit") */
  /* optional per-launch timing of the expand kernel proper (events on the launch stream) */
  static const int kTimingRing = 64;
  bool timing = false;
  rt_event_t ev[2 * kTimingRing];
  uint32_t ev_n = 0;
};

static thread_local std::string g_last_error;

static int fail(int code, const char* what, const char* detail) {
  g_last_error = std::string(what) + (detail ? std::string(": ") + detail : std::string());
  return code;
}
#define RT_CHECK(expr)                                      \
  do {                                                      \
    const char* _e = (expr);                                \
    if (_e) return fail(WGA_E_HIP, #expr, _e);              \
  } while (0)
#define LAUNCH_CHECK()                                      \
  do {                                                      \
    const char* _e = rt_launch_error();                     \
    if (_e) return fail(WGA_E_HIP, "kernel launch", _e);    \
  } while (0)

static int ctx_bind(wga_ctx* c) {
  if (!c) return fail(WGA_E_INVALID_ARG, "null context", nullptr);
  RT_CHECK(rt_set_device(c->device));
  return WGA_OK;
}

/* grow-only scratch arena on the context (scan partials etc.) */
static int ctx_scratch(wga_ctx* c, size_t bytes, void** out) {
  /* whoever takes the scratch may overwrite the two counters the last row-kernel launch left in it: "expand_stream_left_to_v1"
   * and "pseudo_stream_left_to_blocks" then answer "not known" instead of reading someone else's bytes (the launches that own
   * the counters set the pointers again behind this call) */
  c->stream_counts = nullptr;
  c->pseudo_counts = nullptr;
  if (c->scratch_cap < bytes) {
    RT_CHECK(rt_sync(c->stream));
    if (c->scratch) RT_CHECK(rt_free(c->scratch));
    c->scratch = nullptr;
    c->scratch_cap = 0;
    size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
    RT_CHECK(rt_malloc(&c->scratch, cap));
    c->scratch_cap = cap;
  }
  *out = c->scratch;
  return WGA_OK;
}

/* exclusive scan driver shared by wga_exclusive_scan_u64 and the layout */
template <typename F>
static int run_scan_ws(wga_ctx* c, F f, u32 n, u64* d_out /* n+1 */, u64* partial /* n/1024 + 2 */) {
  u32 nb = (n + 1023u) / 1024u;
  if (nb) {
    WGA_LAUNCH(k_scan_partials<F>, nb, WGA_BLOCK, c->stream, f, n, partial);
    LAUNCH_CHECK();
  }
  WGA_LAUNCH(k_scan_top, 1, WGA_BLOCK, c->stream, partial, nb, d_out + n);
  LAUNCH_CHECK();
  if (nb) {
    WGA_LAUNCH(k_scan_final<F>, nb, WGA_BLOCK, c->stream, f, n, (const u64*)partial, d_out);
    LAUNCH_CHECK();
  }
  return WGA_OK;
}
template <typename F>
static int run_scan(wga_ctx* c, F f, u32 n, u64* d_out /* n+1 */) {
  u32 nb = (n + 1023u) / 1024u;
  void* ws;
  int rc = ctx_scratch(c, ((size_t)nb + 1) * sizeof(u64), &ws);
  if (rc) return rc;
  u64* partial = (u64*)ws;
  if (nb) {
    WGA_LAUNCH(k_scan_partials<F>, nb, WGA_BLOCK, c->stream, f, n, partial);
    LAUNCH_CHECK();
  }
  WGA_LAUNCH(k_scan_top, 1, WGA_BLOCK, c->stream, partial, nb, d_out + n);
  LAUNCH_CHECK();
  if (nb) {
    WGA_LAUNCH(k_scan_final<F>, nb, WGA_BLOCK, c->stream, f, n, (const u64*)partial, d_out);
    LAUNCH_CHECK();
  }
  return WGA_OK;
}

/* K11 driver: element sizes -> exclusive scan -> per-record totals (the count call) or the fill.  The count call's scan stays for
 * the fill call of the same protocol (keyed by the entry point, its arrays and the counts, like the piece tables of K7 / K10 /
 * K12): the fill call then is the fill kernel alone — the scan it used to repeat was more than half of it. */
template <typename F>
static int run_elems(wga_ctx* c, int kind, F f, u32 n, uint64_t n_elems, const uint64_t* d_elem_off, uint64_t* d_cnt,
                     typename F::out_t* d_out, const uint64_t* d_out_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (n == 0) return WGA_OK;
  if (!d_elem_off) return fail(WGA_E_INVALID_ARG, "element offsets null", nullptr);
  if (n_elems > 0xFFFFFFF0ull) return fail(WGA_E_INVALID_ARG, "too many elements for one call", nullptr);
  if (!d_out && !d_cnt) return fail(WGA_E_INVALID_ARG, "d_cnt null", nullptr);
  if (d_out && !d_out_off) return fail(WGA_E_INVALID_ARG, "d_out_off null", nullptr);
  const u32 ne = (u32)n_elems;
  static_assert(sizeof(F) <= sizeof(((wga_ctx::ElemScan*)nullptr)->src), "functor larger than the key");
  wga_ctx::ElemScan& es = c->elem_scan;
  unsigned char src[sizeof(es.src)];
  memset(src, 0, sizeof(src));
  memcpy(src, &f, sizeof(F));
  const bool hit = d_out && es.valid && es.kind == kind && es.elem_off == (const void*)d_elem_off && es.n == n && es.ne == ne &&
                   memcmp(es.src, src, sizeof(src)) == 0;
  es.valid = false; /* one shot: a hit is the fill call of the protocol and consumes what the count call left */
  if (!hit) {
    const size_t need = ((size_t)ne + 1 + (size_t)ne / 1024 + 4) * sizeof(u64);
    if (es.cap < need) {
      if (es.mem) RT_CHECK(rt_free(es.mem));
      es.mem = nullptr;
      es.cap = 0;
      RT_CHECK(rt_malloc(&es.mem, need + need / 4));
      es.cap = need + need / 4;
    }
    ScanElem<F> sf;
    sf.f = f;
    sf.elem_off = (const u64*)d_elem_off;
    sf.n = n;
    u64* const esc0 = (u64*)es.mem;
    if ((rc = run_scan_ws(c, sf, ne, esc0, esc0 + ne + 1))) return rc;
    if (!d_out) { /* the count call of the protocol: its scan stays */
      es.kind = kind;
      es.elem_off = (const void*)d_elem_off;
      es.n = n;
      es.ne = ne;
      memcpy(es.src, src, sizeof(src));
      es.valid = true;
    }
  }
  u64* const esc = (u64*)es.mem;
  if (!d_out) {
    WGA_LAUNCH(k_elem_rec_totals, (n + 255u) / 256u, WGA_BLOCK, c->stream, n, (const u64*)d_elem_off,
               (const u64*)esc, (u64*)d_cnt);
    LAUNCH_CHECK();
  } else if (ne) {
    const u32 nb = (ne + 255u) / 256u;
    void* ws;
    if ((rc = ctx_scratch(c, (size_t)nb * sizeof(wga_elem_block), &ws))) return rc;
    WGA_LAUNCH(k_elem_blocks, (nb + 255u) / 256u, WGA_BLOCK, c->stream, n, ne, (const u64*)d_elem_off, (const u64*)esc,
               (const u64*)d_out_off, (wga_elem_block*)ws);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_elem_fill<F>, nb, WGA_BLOCK, c->stream, f, n, ne, (const u64*)d_elem_off, (const u64*)esc, d_out,
               (const u64*)d_out_off, (const wga_elem_block*)ws);
    LAUNCH_CHECK();
  }
  return WGA_OK;
}

static MafRunSrc maf_run_src(const uint64_t* d_runs, const uint64_t* d_run_off, const uint64_t* d_cols) {
  MafRunSrc s;
  s.runs = (const u64*)d_runs;
  s.run_off = (const u64*)d_run_off;
  s.cols = (const u64*)d_cols;
  return s;
}

/* K13 / K14 driver: delimiter lists (count, scan, fill) in the context scratch, then one thread per line.
 * MODE 0 = PAF (wga_paf_line), 1 = MAF (wga_maf_line). */
template <int MODE>
static int split_lines(wga_ctx* c, const uint8_t* d_text, uint64_t n_bytes, uint64_t* n_lines, void* d_lines,
                       uint64_t cap_lines) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!n_lines) return fail(WGA_E_INVALID_ARG, "n_lines null", nullptr);
  *n_lines = 0;
  if (n_bytes == 0) return WGA_OK;
  if (!d_text) return fail(WGA_E_INVALID_ARG, "d_text null", nullptr);
  if (n_bytes >= 0xFFFFFFFFull) return fail(WGA_E_INVALID_ARG, "text of 4 GiB or more: split it at line ends", nullptr);
  const u32 nb = (u32)((n_bytes + 4095u) / 4096u);
  const size_t head = ((size_t)nb + 1 + (size_t)nb / 1024 + 4) * sizeof(u64);
  u64 tot = 0;
  void* ws = nullptr;
  if ((rc = ctx_scratch(c, head, &ws))) return rc;
  for (int attempt = 0; attempt < 2; attempt++) {
    u64* blk = (u64*)c->scratch;
    WGA_LAUNCH((k_paf_delims<false, MODE>), nb, WGA_BLOCK, c->stream, d_text, (u64)n_bytes, blk, (const u64*)nullptr,
               (u64*)nullptr, (u64*)nullptr);
    LAUNCH_CHECK();
    /* exclusive scan of the block counts in place (k_scan_final reads its four values, then writes them) */
    ScanPlain f;
    f.in = blk;
    if ((rc = run_scan_ws(c, f, nb, blk, blk + nb + 1))) return rc;
    RT_CHECK(rt_d2h(&tot, blk + nb, sizeof(u64), c->stream));
    /* the two lists follow the block offsets; their sizes are only known now: growing the arena
     * drops its contents, so the count pass is repeated once */
    const size_t want = head + ((size_t)(tot & 0xFFFFFFFFull) + (size_t)(tot >> 32) + 2) * sizeof(u64);
    if (c->scratch_cap >= want) break;
    if ((rc = ctx_scratch(c, want, &ws))) return rc;
  }
  const u64 n_delims = tot & 0xFFFFFFFFull, n_newlines = tot >> 32;
  u8 last = 0;
  RT_CHECK(rt_d2h(&last, d_text + n_bytes - 1, 1, c->stream));
  *n_lines = n_newlines + (last != (u8)0x0A ? 1 : 0);
  if (!d_lines) return WGA_OK;
  if (cap_lines < *n_lines) return fail(WGA_E_TOO_SMALL, "d_lines too small", nullptr);
  u64* blk_off = (u64*)c->scratch;
  u64* delims = (u64*)((char*)c->scratch + head);
  u64* nl_idx = delims + n_delims + 1;
  WGA_LAUNCH((k_paf_delims<true, MODE>), nb, WGA_BLOCK, c->stream, d_text, (u64)n_bytes, (u64*)nullptr,
             (const u64*)blk_off, delims, nl_idx);
  LAUNCH_CHECK();
  if (MODE == 0) {
    WGA_LAUNCH(k_paf_fields, (u32)((*n_lines + 255u) / 256u), WGA_BLOCK, c->stream, d_text, (u64)n_bytes,
               (u64)*n_lines, n_newlines, n_delims, (const u64*)delims, (const u64*)nl_idx, (wga_paf_line_dev*)d_lines);
  } else {
    WGA_LAUNCH(k_maf_lines, (u32)((*n_lines + 255u) / 256u), WGA_BLOCK, c->stream, d_text, (u64)n_bytes,
               (u64)*n_lines, n_newlines, n_delims, (const u64*)delims, (const u64*)nl_idx, (wga_maf_line_dev*)d_lines);
  }
  LAUNCH_CHECK();
  return WGA_OK;
}

/* K3 / K4: the stream kernel over every block that is not long, then the long blocks piece by piece (wga_k3_maf.h).  Five
 * launches at most, all of them queued whatever the data holds: the table of long blocks is built and sized on the device (its
 * bounds — n list entries, 32 768 + n pieces — are known here), nothing is read back. */
#ifdef WGA_EMU
#define WGA_MAF_PIECE_GRID 3u /* the emulator makes 256 fibers per block, empty or not */
#else
#define WGA_MAF_PIECE_GRID 1024u /* 4 096 resident waves: a wave takes two pieces of a 10^8-column block and adds their counters up before it touches memory */
#endif
template <bool CALLER>
static int maf_walk_call(wga_ctx* c, u32 n, const u8* d_rows, const u64* d_t_off, const u64* d_q_off, const u64* d_cols,
                         const u8* d_strand_neg, wga_cigar_counts* d_counts, u64* d_run_cnt, u64* d_runs,
                         const u64* d_run_off) {
  const size_t cap = (size_t)n + WGA_MAF_PIECE_BUDGET + 1;
  const size_t o_list = 64, o_off = o_list + (((size_t)n * 4 + 63) & ~(size_t)63), o_ptot = o_off + ((((size_t)n + 1) * 4 + 63) & ~(size_t)63),
               o_ex = o_ptot + cap * sizeof(wga_maf_piece_tot), need = o_ex + (cap + 1) * sizeof(wga_maf_piece_tot);
  if (c->maf_tab_cap < need) {
    RT_CHECK(rt_sync(c->stream));
    if (c->maf_tab) RT_CHECK(rt_free(c->maf_tab));
    c->maf_tab = nullptr;
    c->maf_tab_cap = 0;
    RT_CHECK(rt_malloc(&c->maf_tab, need + need / 4));
    c->maf_tab_cap = need + need / 4;
    c->maf_hdr_clean = false;
  }
  if (!c->maf_hdr_clean) { /* a fresh table, or a call that did not get as far as its plan */
    RT_CHECK(rt_memset(c->maf_tab, 0, 64, c->stream));
    c->maf_hdr_clean = true;
  }
  char* const base = (char*)c->maf_tab;
  wga_maf_long_hdr* const hdr = (wga_maf_long_hdr*)base;
  u32* const long_list = (u32*)(base + o_list);
  u32* const list_off = (u32*)(base + o_off);
  wga_maf_piece_tot* const ptot = (wga_maf_piece_tot*)(base + o_ptot);
  wga_maf_piece_tot* const ex = (wga_maf_piece_tot*)(base + o_ex);
  u32 G = c->maf_group ? c->maf_group : n / 24576u; /* eight blocks per wave where that still leaves every CU a few rounds of waves */
  G = G < 1u ? 1u : G > WGA_MAF_G ? WGA_MAF_G : G;
  const u32 grid = (u32)(((u64)n + 4ull * G - 1ull) / (4ull * G));
  /* the fill call of the two-call protocol finds the table its count call built (the long blocks, their pieces and the pieces'
   * totals): it neither lists the long blocks again nor walks them a second time for their totals */
  wga_ctx::MafKey key;
  key.caller = CALLER, key.n = n, key.rows = d_rows, key.t_off = d_t_off, key.q_off = d_q_off, key.cols = d_cols;
  key.long_cols = c->maf_long_cols, key.piece_cols = c->maf_piece_cols;
  const bool hit = d_runs && c->maf_key.valid && c->maf_key.same(key);
  c->maf_key.valid = false;
  if (!hit) c->maf_hdr_clean = false; /* until the plan has cleared the appends */
  if (d_runs)
    WGA_LAUNCH((k_maf_stream<CALLER, true>), grid, WGA_BLOCK, c->stream, n, G, d_rows, d_t_off, d_q_off, d_cols, d_strand_neg, d_counts,
               d_run_cnt, d_runs, d_run_off, (u64)c->maf_long_cols, hit ? (wga_maf_long_hdr*)nullptr : hdr, long_list);
  else
    WGA_LAUNCH((k_maf_stream<CALLER, false>), grid, WGA_BLOCK, c->stream, n, G, d_rows, d_t_off, d_q_off, d_cols, d_strand_neg, d_counts,
               d_run_cnt, d_runs, d_run_off, (u64)c->maf_long_cols, hdr, long_list);
  LAUNCH_CHECK();
  if (!hit) {
    WGA_LAUNCH(k_maf_long_plan, 1, 1024, c->stream, hdr, (const u32*)long_list, list_off, d_cols, (u64)c->maf_piece_cols);
    LAUNCH_CHECK();
    c->maf_hdr_clean = true;
    /* the fill call must not add to what the count call left in the caller's arrays */
    WGA_LAUNCH((k_maf_piece_walk<CALLER, 0>), WGA_MAF_PIECE_GRID, WGA_BLOCK, c->stream, d_rows, d_t_off, d_q_off, d_cols, d_strand_neg,
               (const wga_maf_long_hdr*)hdr, (const u32*)long_list, (const u32*)list_off, ptot, (const wga_maf_piece_tot*)nullptr,
               d_runs ? (wga_cigar_counts*)nullptr : d_counts, d_runs ? (u64*)nullptr : d_run_cnt, (u64*)nullptr, (const u64*)nullptr);
    LAUNCH_CHECK();
  }
  if (!d_runs) {
    c->maf_key = key;
    c->maf_key.valid = true;
    return WGA_OK;
  }
  if (!d_runs) return WGA_OK;
  WGA_LAUNCH(k_maf_piece_scan, 1, 1024, c->stream, (const wga_maf_long_hdr*)hdr, (const wga_maf_piece_tot*)ptot, ex);
  LAUNCH_CHECK();
  WGA_LAUNCH((k_maf_piece_walk<CALLER, 1>), WGA_MAF_PIECE_GRID, WGA_BLOCK, c->stream, d_rows, d_t_off, d_q_off, d_cols, d_strand_neg,
             (const wga_maf_long_hdr*)hdr, (const u32*)long_list, (const u32*)list_off, ptot, (const wga_maf_piece_tot*)ex,
             (wga_cigar_counts*)nullptr, (u64*)nullptr, d_runs, d_run_off);
  LAUNCH_CHECK();
  return WGA_OK;
}

static bool op_all_pieces(const wga_ctx* c, const wga_cigar_batch* b) {
  return b->n && b->n_ops > c->op_long_ops && b->n_ops / b->n > c->op_long_ops / 2;
}
/* the piece table of the op walks whose records can be long (K7, K10, K12): per record the number of pieces (0: the one-wave
 * kernel keeps it), their exclusive scan, every piece's record and `per_piece` bytes per piece, in c->op_tab.  Nothing comes
 * back to the host: the table is sized by a bound (a record of nops > long_ops ops has at most nops / piece_ops + 1 pieces,
 * and at most n_ops / long_ops records are long), t.np is that bound (0 when no record can be long) and the walks read the
 * number of pieces from piece_off[n].  With `reuse` and a table built under the same key nothing is launched (the fill
 * call of the protocol); otherwise the table is rebuilt and left invalid — the caller validates it (op_tab_keep) once its
 * count walk and record scan are queued. */
static int op_piece_table(wga_ctx* c, const wga_cigar_batch* b, size_t per_piece, const wga_ctx::OpTabKey& key, bool reuse,
                          bool* hit) {
  wga_ctx::OpTab& t = c->op_tab;
  *hit = reuse && t.valid && t.key == key;
  t.valid = false; /* one shot: the fill call that takes the table consumes it (its arrays stay where they are for this call) */
  if (*hit) return WGA_OK;
  t.np = 0;
  t.all = false;
  if (b->n_ops <= c->op_long_ops) return WGA_OK;
  const u32 n = b->n;
  int rc;
  /* a batch of mostly long records: the few short ones are one piece each, so that one grid walks everything (the one-wave
   * kernel would run for the length of its longest record with the chip nearly empty) */
  t.all = op_all_pieces(c, b);
  const u64 n_long = t.all ? (u64)n : (b->n_ops / c->op_long_ops < (u64)n ? b->n_ops / c->op_long_ops : (u64)n);
  const u64 bound = b->n_ops / c->op_piece_ops + n_long + 1;
  if (bound > 0xFFFFFFF0ull) return fail(WGA_E_INVALID_ARG, "too many pieces for one call", nullptr);
  const u32 np = (u32)bound;
  const size_t head = (((size_t)n * 2 + 2 + (size_t)n / 1024 + 4) * sizeof(u64) + 63) & ~(size_t)63;
  const size_t want = head + (((size_t)np * per_piece + 63) & ~(size_t)63) + (size_t)np * sizeof(u32) + 64;
  if (t.cap < want) {
    RT_CHECK(rt_sync(c->stream));
    if (t.mem) RT_CHECK(rt_free(t.mem));
    t.mem = nullptr, t.cap = 0;
    const size_t cap = want < (1u << 20) ? (1u << 20) : want + want / 2;
    RT_CHECK(rt_malloc(&t.mem, cap));
    t.cap = cap;
  }
  u64* npieces = (u64*)t.mem;
  u64* off = npieces + n;
  u64* partial = off + n + 1;
  WGA_LAUNCH(k_op_piece_counts, (n + 255u) / 256u, WGA_BLOCK, c->stream, n, (const u64*)b->d_op_off, (u64)c->op_long_ops,
             (u64)c->op_piece_ops, (u32)t.all, npieces);
  LAUNCH_CHECK();
  ScanPlain sp;
  sp.in = npieces;
  if ((rc = run_scan_ws(c, sp, n, off, partial))) return rc;
  t.np = np;
  t.piece_off = off;
  t.pieces = (char*)t.mem + head;
  t.piece_rec = (u32*)((char*)t.pieces + (((size_t)np * per_piece + 63) & ~(size_t)63));
  WGA_LAUNCH(k_op_piece_records, (n + 255u) / 256u, WGA_BLOCK, c->stream, n, (const u64*)t.piece_off, t.piece_rec);
  LAUNCH_CHECK();
  return WGA_OK;
}
static void op_tab_keep(wga_ctx* c, const wga_ctx::OpTabKey& key) {
  c->op_tab.key = key;
  c->op_tab.valid = true;
}

extern "C" {

int wga_abi_version(void) { return WGA_ABI_VERSION; }
const char* wga_last_error(void) { return g_last_error.c_str(); }
int wga_device_count(void) { return rt_device_count(); }

int wga_ctx_create(int device, wga_ctx** out) {
  if (!out) return fail(WGA_E_INVALID_ARG, "out is null", nullptr);
  int n = rt_device_count();
  if (n <= 0) return fail(WGA_E_NO_DEVICE, "no HIP device visible (libwgahip needs an MI355X)", nullptr);
  if (device < 0 || device >= n) return fail(WGA_E_INVALID_ARG, "device index out of range", nullptr);
  wga_ctx* c = new wga_ctx();
  c->device = device;
  const char* e = rt_set_device(device);
  if (!e) e = rt_stream_create(&c->own_stream);
  if (e) {
    delete c;
    return fail(WGA_E_HIP, "context creation", e);
  }
  c->stream = c->own_stream;
  /* A/B switch for measurements: WGA_EXPAND_VARIANT=0 selects v1 of the paf2maf row kernel (wga_ctx_set_param overrides) */
  if (const char* v = getenv("WGA_EXPAND_VARIANT")) c->expand_variant = (atoi(v) == 0 || atoi(v) == 3) ? atoi(v) : -1;
  if (const char* v = getenv("WGA_EXPAND_DRAIN_MIN")) {
    const int d = atoi(v);
    if (d >= 0 && d <= 64) c->expand_drain_min = (unsigned)d;
  }
  *out = c;
  return WGA_OK;
}

void wga_ctx_destroy(wga_ctx* c) {
  if (!c) return;
  (void)rt_set_device(c->device);
  (void)rt_sync(c->stream);
  if (c->timing)
    for (int k = 0; k < 2 * wga_ctx::kTimingRing; k++) rt_event_destroy(c->ev[k]);
  if (c->rs_have_ev) rt_event_destroy(c->rs_ready), rt_event_destroy(c->rs_done);
  for (rt_event_t e : c->rs_copied) rt_event_destroy(e);
  for (wga_stream_t st : c->rs_streams) rt_stream_destroy(st);
  if (c->scratch) (void)rt_free(c->scratch);
  if (c->elem_scan.mem) (void)rt_free(c->elem_scan.mem);
  if (c->class_tab.mem) (void)rt_free(c->class_tab.mem);
  if (c->cov_pieces) (void)rt_free(c->cov_pieces);
  if (c->cov_list) (void)rt_free(c->cov_list);
  if (c->cov_order) (void)rt_free(c->cov_order);
  if (c->cov_tile_list) (void)rt_free(c->cov_tile_list);
  if (c->op_tab.mem) (void)rt_free(c->op_tab.mem);
  if (c->maf_tab) (void)rt_free(c->maf_tab);
  rt_stream_destroy(c->own_stream);
  delete c;
}

/* The context's scratch arenas are ordered on ONE stream: work still in flight on the stream that is being left must
 * not see them reused or regrown by calls on the new one, so a switch drains the old stream first. */
int wga_ctx_set_stream(wga_ctx* c, void* hip_stream) {
  int rc = ctx_bind(c);
  if (rc) return rc;
#ifdef WGA_EMU
  (void)hip_stream; /* a caller's stream handle means nothing to the emulator: everything runs in call order anyway */
  return WGA_OK;
#else
  if (c->stream != (wga_stream_t)hip_stream) RT_CHECK(rt_sync(c->stream));
  c->stream = (wga_stream_t)hip_stream;
  return WGA_OK;
#endif
}

int wga_ctx_reset_stream(wga_ctx* c) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (c->stream != c->own_stream) RT_CHECK(rt_sync(c->stream));
  c->stream = c->own_stream;
  return WGA_OK;
}

int wga_ctx_set_param(wga_ctx* c, const char* name, int64_t value) {
  if (!c || !name) return fail(WGA_E_INVALID_ARG, "null argument", nullptr);
  if (strcmp(name, "expand_force_slow") == 0) {
    c->expand_force_slow = value != 0;
    return WGA_OK;
  }
  if (strcmp(name, "reduce_same_device_ok") == 0) { /* wga_reduce_scatter_i32 over contexts that share a device (tests) */
    c->rs_same_device_ok = value != 0;
    return WGA_OK;
  }
  if (strcmp(name, "cov_spin_limit") == 0) { /* K5's list pass: polls of a tile sum before the look-back adds up the ops itself */
    if (value < 0 || value > 0x7FFFFFFF) return fail(WGA_E_INVALID_ARG, "cov_spin_limit: 0 .. 2^31 - 1", nullptr);
    c->cov_spin_limit = (u32)value;
    return WGA_OK;
  }
  if (strcmp(name, "reduce_staged") == 0) { /* wga_reduce_scatter_i32 by staged peer copies even where peer access exists */
    c->rs_staged = value != 0;
    return WGA_OK;
  }
  if (strcmp(name, "expand_drain_min") == 0) { /* 0 = chosen by the size of the sequence pools (WGA_DRAIN_POOL_BYTES) */
    if (value < 0 || value > 64) return fail(WGA_E_INVALID_ARG, "expand_drain_min: 0 .. 64", nullptr);
    c->expand_drain_min = (unsigned)value;
    return WGA_OK;
  }
  if (strcmp(name, "expand_no_table") == 0) {
    c->expand_no_table = value != 0;
    return WGA_OK;
  }
  if (strcmp(name, "op_long_ops") == 0) {
    if (value < 1) return fail(WGA_E_INVALID_ARG, "must be positive", name);
    c->op_long_ops = (uint64_t)value;
    return WGA_OK;
  }
  if (strcmp(name, "op_piece_ops") == 0) {
    if (value < 256 || (value & 255)) return fail(WGA_E_INVALID_ARG, "a positive multiple of 256", name);
    c->op_piece_ops = (uint64_t)value;
    return WGA_OK;
  }
  if (strcmp(name, "maf_long_cols") == 0 || strcmp(name, "maf_piece_cols") == 0) {
    if (value < 1) return fail(WGA_E_INVALID_ARG, "must be positive", name);
    (name[4] == 'l' ? c->maf_long_cols : c->maf_piece_cols) = (uint64_t)value;
    return WGA_OK;
  }
  if (strcmp(name, "maf_group") == 0) {
    if (value < 0 || value > (long long)WGA_MAF_G) return fail(WGA_E_INVALID_ARG, "maf_group: 0 (by the batch) .. 8", nullptr);
    c->maf_group = (unsigned)value;
    return WGA_OK;
  }
  if (strcmp(name, "expand_variant") == 0) {
    if (value != -1 && value != 0 && value != 3) return fail(WGA_E_INVALID_ARG, "expand_variant: -1 (the library's choice), 0, 3", nullptr);
    c->expand_variant = (int)value;
    return WGA_OK;
  }
  if (strcmp(name, "expand_job_tiles") == 0) { /* streaming row kernel: consecutive tiles per wave */
    if (value < 0 || value > (int64_t)WGA_S_MAX_JOB_TILES) return fail(WGA_E_INVALID_ARG, "expand_job_tiles: 0 (by the batch), 1 .. 32", nullptr);
    c->expand_job_tiles = (int)value;
    return WGA_OK;
  }
  if (strcmp(name, "pseudo_variant") == 0) { /* wga_pafpseudo_fill: 3 the streaming row kernel (default), 0 one block per tile */
    if (value != 0 && value != 3) return fail(WGA_E_INVALID_ARG, "pseudo_variant: 0, 3", nullptr);
    c->pseudo_variant = (int)value;
    return WGA_OK;
  }
  if (strcmp(name, "expand_timing") == 0) {
    if (value && !c->timing) {
      int rc = ctx_bind(c);
      if (rc) return rc;
      for (int k = 0; k < 2 * wga_ctx::kTimingRing; k++) {
        const char* e = rt_event_create(&c->ev[k]);
        if (e) { /* give back what was created: `timing` stays off, nobody else would destroy them */
          for (int j = 0; j < k; j++) rt_event_destroy(c->ev[j]);
          return fail(WGA_E_HIP, "rt_event_create", e);
        }
      }
    }
    if (!value && c->timing)
      for (int k = 0; k < 2 * wga_ctx::kTimingRing; k++) rt_event_destroy(c->ev[k]);
    c->timing = value != 0;
    c->ev_n = 0;
    return WGA_OK;
  }
  return fail(WGA_E_INVALID_ARG, "unknown parameter", name);
}

int wga_sync(wga_ctx* c) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  RT_CHECK(rt_sync(c->stream));
  return WGA_OK;
}
int wga_malloc(wga_ctx* c, size_t bytes, void** d_out) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!d_out) return fail(WGA_E_INVALID_ARG, "d_out is null", nullptr);
  const char* e = rt_malloc(d_out, bytes);
  if (e) return fail(WGA_E_OOM, "device allocation", e);
  return WGA_OK;
}
/* What a count call left for its fill call (the K11 scan, the piece table of K7 / K10 / K12, pafpseudo's class sums) is keyed
 * by the arrays it was made from.  Writing into one of those arrays through the library, or freeing it, drops it: a fill call
 * then computes its own.  [lo, lo + bytes) is the range written (bytes == 0: the allocation that starts at lo). */
static void ctx_arrays_written(wga_ctx* c, const void* lo, size_t bytes) {
  const uintptr_t a = (uintptr_t)lo, z = a + (bytes ? bytes : 1);
  /* an array whose extent the key holds (the packed ops, the CSR offsets) is hit by a write anywhere inside it — an upload at an
   * offset, a rewritten tail —, the others (per-record arrays of an entry point's own layout) by a write that covers their start */
  auto in = [&](const void* q) { return q && (uintptr_t)q >= a && (uintptr_t)q < z; };
  auto hits = [&](const void* q, size_t extent) { return q && (uintptr_t)q < z && (uintptr_t)q + (extent ? extent : 1) > a; };
  const wga_ctx::OpTabKey& k = c->op_tab.key;
  if (hits(k.ops, (size_t)k.n_ops * 4) || hits(k.op_off, ((size_t)k.n + 1) * 8) || in(k.x0) || in(k.x1) || in(k.x2)) c->op_tab.valid = false;
  wga_ctx::ElemScan& es = c->elem_scan;
  const void* src[sizeof(es.src) / sizeof(void*)];
  memcpy(src, es.src, sizeof(es.src));
  for (const void* q : src)
    if (in(q)) es.valid = false;
  if (hits(es.elem_off, ((size_t)es.n + 1) * 8)) es.valid = false;
  const wga_ctx::MafKey& mk = c->maf_key;
  if (in(mk.rows) || hits(mk.t_off, (size_t)mk.n * 8) || hits(mk.q_off, (size_t)mk.n * 8) || hits(mk.cols, (size_t)mk.n * 8)) c->maf_key.valid = false;
  if (hits(c->class_tab.ops, (size_t)c->class_tab.n_ops * 4) || hits(c->class_tab.op_off, ((size_t)c->class_tab.n + 1) * 8)) c->class_tab.valid = false;
}

int wga_free(wga_ctx* c, void* d_ptr) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (d_ptr) ctx_arrays_written(c, d_ptr, 0); /* what a count call left for its fill call does not outlive the arrays it was made from */
  if (d_ptr) RT_CHECK(rt_free(d_ptr));
  return WGA_OK;
}
int wga_memcpy_h2d(wga_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (bytes) {
    ctx_arrays_written(c, d_dst, bytes);
    RT_CHECK(rt_h2d(d_dst, h_src, bytes, c->stream));
  }
  return WGA_OK;
}
int wga_memcpy_d2h(wga_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (bytes)
    RT_CHECK(rt_d2h(h_dst, d_src, bytes, c->stream));
  else
    RT_CHECK(rt_sync(c->stream));
  return WGA_OK;
}
int wga_host_alloc(wga_ctx* c, size_t bytes, void** h_out) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!h_out) return fail(WGA_E_INVALID_ARG, "h_out is null", nullptr);
  const char* e = rt_host_alloc(h_out, bytes);
  if (e) return fail(WGA_E_OOM, "pinned host allocation", e);
  return WGA_OK;
}
int wga_host_free(wga_ctx* c, void* h_ptr) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (h_ptr) RT_CHECK(rt_host_free(h_ptr));
  return WGA_OK;
}
int wga_memcpy_d2h_async(wga_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (bytes) RT_CHECK(rt_d2h_async(h_dst, d_src, bytes, c->stream));
  return WGA_OK;
}
int wga_memset(wga_ctx* c, void* d_dst, int byte, size_t bytes) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (bytes) {
    ctx_arrays_written(c, d_dst, bytes);
    RT_CHECK(rt_memset(d_dst, byte, bytes, c->stream));
  }
  return WGA_OK;
}

/* ------------------------------------------------------------------------------------------ */
static int check_batch(const wga_cigar_batch* b) {
  if (!b) return fail(WGA_E_INVALID_ARG, "batch is null", nullptr);
  if (b->n && (!b->d_op_off || !b->d_strand_neg)) return fail(WGA_E_INVALID_ARG, "batch arrays null", nullptr);
  if (b->n_ops && !b->d_ops) return fail(WGA_E_INVALID_ARG, "d_ops null", nullptr);
  if (((uintptr_t)b->d_ops & 15u) != 0) return fail(WGA_E_INVALID_ARG, "d_ops must be 16-byte aligned", nullptr);
  return WGA_OK;
}
static inline u64 n_tiles(u64 n_ops) { return (n_ops + WGA_TILE - 1) / WGA_TILE; }

size_t wga_tile_ws_bytes(uint64_t n_ops) { return (size_t)(n_tiles(n_ops) * sizeof(wga_tile_sum)) + 16; }

int wga_cigar_stat(wga_ctx* c, const wga_cigar_batch* b, wga_cigar_counts* d_counts,
                   wga_rec_diag* d_diag, void* d_tile_ws) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if ((rc = check_batch(b))) return rc;
  if (b->n == 0) return WGA_OK;
  if (!d_counts || !d_diag) return fail(WGA_E_INVALID_ARG, "d_counts / d_diag null", nullptr);
  RT_CHECK(rt_memset(d_counts, 0, (size_t)b->n * sizeof(wga_cigar_counts), c->stream));
  RT_CHECK(rt_memset(d_diag, 0xFF, (size_t)b->n * sizeof(wga_rec_diag), c->stream));
  u64 nt = n_tiles(b->n_ops);
  if (nt == 0) return WGA_OK;
  void* ws;
  if ((rc = ctx_scratch(c, (size_t)nt * sizeof(wga_tile_rec), &ws))) return rc;
  wga_tile_rec* tile_rec = (wga_tile_rec*)ws;
  WGA_LAUNCH(k_tile_rec, (u32)((nt + 255) / 256), WGA_BLOCK, c->stream, (const u64*)b->d_op_off,
             b->d_strand_neg, b->n, (u64)b->n_ops, tile_rec);
  LAUNCH_CHECK();
  const u32 grid = (u32)((nt + 3) / 4);
  WGA_LAUNCH(k_cigar_stat, grid, WGA_BLOCK, c->stream, b->d_ops, (const u64*)b->d_op_off,
             b->d_strand_neg, b->n, (u64)b->n_ops, (const wga_tile_rec*)tile_rec, d_counts, d_diag,
             (wga_tile_sum*)d_tile_ws);
  LAUNCH_CHECK();
  return WGA_OK;
}

int wga_reduce_scatter_i32(wga_ctx** ctxs, int ngpu, int32_t** d_bufs, uint64_t count) {
  if (!ctxs || !d_bufs || ngpu < 1) return fail(WGA_E_INVALID_ARG, "null argument", nullptr);
  for (int g = 0; g < ngpu; g++) {
    if (!ctxs[g] || (count && !d_bufs[g])) return fail(WGA_E_INVALID_ARG, "null context / buffer", nullptr);
    for (int h = 0; h < g; h++)
      if (ctxs[h] == ctxs[g] || (!ctxs[0]->rs_same_device_ok && ctxs[h]->device == ctxs[g]->device))
        return fail(WGA_E_INVALID_ARG, "two contexts on one device", nullptr);
  }
  if (ngpu == 1 || count == 0) return WGA_OK;
  int rc;
  /* Nothing here waits on the host.  (1) every context records "my buffer is as my stream leaves it"; (2) device g's stream
   * waits for the others' records and adds their slices g to its own — read where they lie, by ONE kernel that has a load
   * per peer in flight in every thread (all of the device's xGMI links carry data at once), or, without peer access, pulled
   * into scratch by N-1 copies on N-1 streams of their own (in flight together as well) and added by the same kernel;
   * (3) every context's stream waits until the others have read its buffer.  Work enqueued behind the call on any of the
   * contexts' streams sees the result. */
  for (int g = 0; g < ngpu; g++) {
    wga_ctx* c = ctxs[g];
    if ((rc = ctx_bind(c))) return rc;
    if (!c->rs_have_ev) {
      RT_CHECK(rt_event_create(&c->rs_ready));
      RT_CHECK(rt_event_create(&c->rs_done));
      c->rs_have_ev = true;
    }
    RT_CHECK(rt_event_record(c->rs_ready, c->stream));
  }
  bool direct = true;
  for (int g = 0; g < ngpu && direct; g++) {
    if (ctxs[g]->rs_staged) direct = false;
    for (int h = 0; h < ngpu && direct; h++)
      if (h != g && rt_peer_enable(ctxs[g]->device, ctxs[h]->device)) direct = false;
  }
  for (int g = 0; g < ngpu; g++) {
    wga_ctx* c = ctxs[g];
    const u64 lo = count * (u64)g / (u64)ngpu, hi = count * (u64)(g + 1) / (u64)ngpu, n = hi - lo;
    if ((rc = ctx_bind(c))) return rc;
    if (n) {
      int* stage = nullptr;
      if (!direct) {
        void* ws;
        if ((rc = ctx_scratch(c, (size_t)n * 4 * (size_t)(ngpu - 1), &ws))) return rc;
        stage = (int*)ws;
        while ((int)c->rs_streams.size() < ngpu - 1) {
          wga_stream_t st;
          rt_event_t ev;
          RT_CHECK(rt_stream_create(&st));
          c->rs_streams.push_back(st);
          RT_CHECK(rt_event_create(&ev));
          c->rs_copied.push_back(ev);
        }
      }
      const u32 grid = (u32)(n / 1024u < 16384u ? (n + 1023u) / 1024u : 16384u);
      int k = 0;
      wga_peer_srcs srcs;
      int n_src = 0;
      auto add = [&]() {
        WGA_LAUNCH(k_add_peers_i32, grid, WGA_BLOCK, c->stream, (int*)d_bufs[g] + lo, srcs, n_src, (u64)n);
        n_src = 0;
      };
      for (int h = 0; h < ngpu; h++) {
        if (h == g) continue;
        if (direct) {
          RT_CHECK(rt_stream_wait_event(c->stream, ctxs[h]->rs_ready));
          srcs.p[n_src++] = (const int*)d_bufs[h] + lo;
        } else { /* the scratch is this stream's: the pull starts behind what the stream had in flight, on a stream of its own */
          wga_stream_t st = c->rs_streams[k];
          RT_CHECK(rt_stream_wait_event(st, c->rs_ready));
          RT_CHECK(rt_stream_wait_event(st, ctxs[h]->rs_ready));
          RT_CHECK(rt_peer_copy(stage + (size_t)k * n, c->device, d_bufs[h] + lo, ctxs[h]->device, (size_t)n * 4, st));
          RT_CHECK(rt_event_record(c->rs_copied[k], st)); /* the staged pieces are named below, WGA_PEER_MAX per launch */
        }
        k++;
        if (n_src == WGA_PEER_MAX && direct) { /* more peers than one launch takes (never on one node) */
          add();
          LAUNCH_CHECK();
        }
      }
      if (!direct) {
        /* every pull has been enqueued — they run side by side — and only now does the adding stream wait for them */
        for (int j = 0; j < k; j++) RT_CHECK(rt_stream_wait_event(c->stream, c->rs_copied[j]));
        for (int j0 = 0; j0 < k; j0 += WGA_PEER_MAX) {
          n_src = 0;
          for (int j = j0; j < k && j < j0 + WGA_PEER_MAX; j++) srcs.p[n_src++] = stage + (size_t)j * n;
          add();
          LAUNCH_CHECK();
        }
      } else if (n_src) {
        add();
        LAUNCH_CHECK();
      }
    }
    RT_CHECK(rt_event_record(c->rs_done, c->stream));
  }
  for (int h = 0; h < ngpu; h++) {
    if ((rc = ctx_bind(ctxs[h]))) return rc;
    for (int g = 0; g < ngpu; g++)
      if (g != h) RT_CHECK(rt_stream_wait_event(ctxs[h]->stream, ctxs[g]->rs_done));
  }
  return WGA_OK;
}

#ifdef WGA_EMU
/* test hook of the emulator build only (not part of the ABI): the most peer copies that were outstanding towards `device` at
 * one time, as the emulator's streams keep the book (wga_rt.h) */
int wga_emu_peer_copies_in_flight(int device) { return emu_book().most[device & 63]; }
void wga_emu_peer_copies_reset(void) { emu_book() = emu_peer_book(); }
#endif

int wga_exclusive_scan_u64(wga_ctx* c, uint32_t n, const uint64_t* d_in, uint64_t* d_out) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!d_out || (n && !d_in)) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  ScanPlain f;
  f.in = (const u64*)d_in;
  return run_scan(c, f, n, (u64*)d_out);
}

int wga_paf2maf_layout(wga_ctx* c, uint32_t n, const wga_cigar_counts* d_counts,
                       const uint64_t* d_t_src_len, const uint64_t* d_q_src_len,
                       const uint32_t* d_pre_t, const uint32_t* d_pre_q, const uint32_t* d_post,
                       uint64_t* d_t_row_off, uint64_t* d_q_row_off, uint64_t* d_rec_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!d_rec_off) return fail(WGA_E_INVALID_ARG, "d_rec_off null", nullptr);
  if (n && (!d_counts || !d_t_src_len || !d_q_src_len || !d_t_row_off || !d_q_row_off))
    return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  ScanLayout f;
  f.counts = d_counts;
  f.t_src_len = (const u64*)d_t_src_len;
  f.q_src_len = (const u64*)d_q_src_len;
  f.pre_t = d_pre_t;
  f.pre_q = d_pre_q;
  f.post = d_post;
  rc = run_scan(c, f, n, (u64*)d_rec_off);
  if (rc) return rc;
  if (n) {
    WGA_LAUNCH(k_layout_rows, (n + 255u) / 256u, WGA_BLOCK, c->stream, f, n, (const u64*)d_rec_off,
               (u64*)d_t_row_off, (u64*)d_q_row_off);
    LAUNCH_CHECK();
  }
  return WGA_OK;
}

/* Tiles per job of the streaming row kernel: eight on a full-size batch; four when the whole grid is only a few rounds of the
 * device's resident waves (an eighth of configs[1] — a rank's share at 8 GPUs: 0.792 against 0.810 ms; full size: the same) */
static inline u32 job_tiles_for(int param, u64 nt) {
  if (param >= 1) return param > (int)WGA_S_MAX_JOB_TILES ? WGA_S_MAX_JOB_TILES : (u32)param;
  return nt < 200000ull ? 4u : 8u;
}

int wga_paf2maf_expand(wga_ctx* c, const wga_cigar_batch* b, const wga_cigar_counts* d_counts,
                       const void* d_tile_ws, const uint8_t* d_t_fa, uint64_t t_fa_bytes,
                       const uint64_t* d_t_src_off, const uint64_t* d_t_src_len,
                       const uint8_t* d_q_fa, uint64_t q_fa_bytes, const uint64_t* d_q_src_off,
                       const uint64_t* d_q_src_len, uint8_t* d_out, const uint64_t* d_t_row_off,
                       const uint64_t* d_q_row_off, wga_rec_diag* d_diag) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if ((rc = check_batch(b))) return rc;
  if (b->n == 0 || b->n_ops == 0) return WGA_OK;
  if (!d_counts || !d_tile_ws || !d_t_src_off || !d_t_src_len || !d_q_src_off || !d_q_src_len ||
      !d_out || !d_t_row_off || !d_q_row_off || !d_diag)
    return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  if ((t_fa_bytes && !d_t_fa) || (q_fa_bytes && !d_q_fa)) return fail(WGA_E_INVALID_ARG, "null sequence pool", nullptr);
  u64 nt = n_tiles(b->n_ops);
  if (nt > 0x7FFFFFFFull) return fail(WGA_E_INVALID_ARG, "batch too large for one launch", nullptr);
  /* pre-pass: per-record descriptors and per-tile base sums, in the context's scratch arena */
  void* ws;
  size_t rec_bytes = ((size_t)b->n * sizeof(wga_rec_desc) + 255) & ~(size_t)255;
  const size_t desc_bytes = (size_t)nt * sizeof(wga_tile_desc);
  const size_t list_bytes = 256 + 2 * (size_t)nt * sizeof(u32); /* two counters + the lists of wide / huge tiles */
  const int variant = c->expand_variant >= 0 ? c->expand_variant : WGA_AUTO_LONG_VARIANT;
  c->expand_variant_used = variant;
  const size_t plan_bytes = 0;
  const size_t flag_bytes = variant == 3 ? (((size_t)nt + 255) & ~(size_t)255) : 0; /* streaming kernel: one byte per tile */
  if ((rc = ctx_scratch(c, rec_bytes + desc_bytes + list_bytes + plan_bytes + flag_bytes, &ws))) return rc;
  wga_rec_desc* recs = (wga_rec_desc*)ws;
  wga_tile_desc* tdesc = (wga_tile_desc*)((char*)ws + rec_bytes);
  u32* const wide_counts = (u32*)((char*)ws + rec_bytes + desc_bytes);
  u32* const wide_list = wide_counts + 64;
  WGA_LAUNCH(k_rec_desc, (b->n + 255u) / 256u, WGA_BLOCK, c->stream, b->n, d_counts,
             b->d_strand_neg, (const u64*)d_t_src_off, (const u64*)d_t_src_len,
             (const u64*)d_q_src_off, (const u64*)d_q_src_len, (const u64*)d_t_row_off,
             (const u64*)d_q_row_off, recs);
  LAUNCH_CHECK();
  WGA_LAUNCH(k_tile_base, (u32)((nt + 255) / 256), WGA_BLOCK, c->stream, (const u64*)b->d_op_off,
             (u64)b->n_ops, (const wga_tile_sum*)d_tile_ws, (const wga_rec_desc*)recs, tdesc, 0);
  LAUNCH_CHECK();
  ExpandArgs a;
  a.ops = b->d_ops;
  a.op_off = (const u64*)b->d_op_off;
  a.n_ops = b->n_ops;
  a.tdesc = tdesc;
  a.recs = recs;
  a.t_fa = d_t_fa;
  a.t_fa_bytes = t_fa_bytes;
  a.q_fa = d_q_fa;
  a.q_fa_bytes = q_fa_bytes;
  a.out = d_out;
  a.diag = d_diag;
  a.force_slow = c->expand_force_slow;
  a.no_table = c->expand_no_table;
  a.drain_min = 0; /* below */
  a.tile_count = nullptr;
  a.tile_list = nullptr;
  a.n_rec = b->n;
  a.job_tiles = job_tiles_for(c->expand_job_tiles, nt);
  const bool stream = variant == 3;
  /* when the gap-touching chunks are emitted (RowSrc::drain_min) */
  /* when v1's waves emit their queued gap-touching chunks (RowSrc::drain_min) */
  a.drain_min = c->expand_drain_min ? c->expand_drain_min : ((u64)t_fa_bytes + (u64)q_fa_bytes > WGA_DRAIN_POOL_BYTES ? 16u : 32u);
  c->expand_drain_min_used = a.drain_min;
  u32* const fast_list = wide_list + nt; /* the second half of the list area: tiles for v1's row emitters */
  if (stream) { /* part of the pre-pass: the tiles the streaming kernel leaves to v1 (records that are not clean, giant tiles) */
    u8* const tile_flag = (u8*)ws + rec_bytes + desc_bytes + list_bytes + plan_bytes;
    RT_CHECK(rt_memset(wide_counts, 0, 256, c->stream));
    RT_CHECK(rt_memset(tile_flag, 0, flag_bytes, c->stream));
    WGA_LAUNCH(k_stream_mark_rec, (b->n + 255u) / 256u, WGA_BLOCK, c->stream, b->n, (const wga_rec_desc*)recs,
               (const u64*)b->d_op_off, (u64)t_fa_bytes, (u64)q_fa_bytes, tile_flag);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_stream_mark_tile, (u32)((nt + 255) / 256), WGA_BLOCK, c->stream, tdesc, (u64)nt, (const u8*)tile_flag,
               c->expand_force_slow, wide_counts, fast_list, wide_list);
    LAUNCH_CHECK();
    c->stream_counts = wide_counts;
  }
  const uint32_t slot = c->ev_n % (uint32_t)wga_ctx::kTimingRing;
  if (c->timing) RT_CHECK(rt_event_record(c->ev[2 * slot], c->stream));
  if (stream) {
    const u64 jobs = (nt + a.job_tiles - 1) / a.job_tiles;
    WGA_LAUNCH(k_paf2maf_expand_s, (u32)jobs, 128u, c->stream, a);
    LAUNCH_CHECK();
    const u32 side_grid = nt < 256 ? (u32)nt : 256u;
    a.tile_count = wide_counts; /* tiles of records that are not clean, tiles beyond 2^24 columns: v1's row emitters */
    a.tile_list = fast_list;
    WGA_LAUNCH(k_paf2maf_expand_list, side_grid, WGA_BLOCK, c->stream, a);
    LAUNCH_CHECK();
    a.force_slow = 1; /* beyond 2^31 columns (and everything under "expand_force_slow"): the op-serial walk */
    a.tile_count = wide_counts + 1;
    a.tile_list = wide_list;
    WGA_LAUNCH(k_paf2maf_expand_list, side_grid, WGA_BLOCK, c->stream, a);
    LAUNCH_CHECK();
  } else {
    WGA_LAUNCH(k_paf2maf_expand, (u32)nt, WGA_BLOCK, c->stream, a);
    LAUNCH_CHECK();
  }
  if (c->timing) {
    RT_CHECK(rt_event_record(c->ev[2 * slot + 1], c->stream));
    c->ev_n++;
  }
  return WGA_OK;
}

int wga_ctx_get_param(wga_ctx* c, const char* name, int64_t* value) {
  if (!c || !name || !value) return fail(WGA_E_INVALID_ARG, "null argument", nullptr);
  if (strcmp(name, "cov_spin_limit") == 0) {
    *value = (int64_t)c->cov_spin_limit;
    return WGA_OK;
  }
  if (strcmp(name, "expand_drain_min") == 0) { /* what the last wga_paf2maf_expand used */
    *value = (int64_t)c->expand_drain_min_used;
    return WGA_OK;
  }
  if (strcmp(name, "expand_variant") == 0) {
    *value = (int64_t)c->expand_variant;
    return WGA_OK;
  }
  if (strcmp(name, "expand_stream_left_to_v1") == 0) { /* tiles the streaming kernel's last launch left to v1 (a device read: diagnostics) */
    *value = c->expand_variant_used == 3 ? -1 : 0; /* -1: the scratch that held the counters has been handed on */
    if (c->expand_variant_used == 3 && c->stream_counts) {
      u32 h[2] = {0, 0};
      int rc = ctx_bind(c);
      if (rc) return rc;
      RT_CHECK(rt_d2h(h, c->stream_counts, sizeof(h), c->stream));
      *value = (int64_t)h[0] + (int64_t)h[1];
    }
    return WGA_OK;
  }
  if (strcmp(name, "expand_job_tiles") == 0) {
    *value = (int64_t)c->expand_job_tiles;
    return WGA_OK;
  }
  if (strcmp(name, "pseudo_variant") == 0) {
    *value = (int64_t)c->pseudo_variant;
    return WGA_OK;
  }
  if (strcmp(name, "pseudo_stream_left_to_blocks") == 0) { /* tiles the last base-mode wga_pafpseudo_fill left to the block kernel */
    *value = -1; /* not known (no such launch yet, or the scratch that held the counters has been handed on) */
    if (c->pseudo_counts) {
      u32 h[2] = {0, 0};
      int rc = ctx_bind(c);
      if (rc) return rc;
      RT_CHECK(rt_d2h(h, c->pseudo_counts, sizeof(h), c->stream));
      *value = (int64_t)h[0] + (int64_t)h[1];
    }
    return WGA_OK;
  }
  if (strcmp(name, "expand_variant_used") == 0) { /* what the last wga_paf2maf_expand ran */
    *value = (int64_t)c->expand_variant_used;
    return WGA_OK;
  }
  return fail(WGA_E_INVALID_ARG, "unknown parameter", name);
}

int wga_ctx_expand_timing(wga_ctx* c, double* ms_sum, uint32_t* launches) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!ms_sum || !launches) return fail(WGA_E_INVALID_ARG, "null argument", nullptr);
  *ms_sum = 0.0;
  *launches = 0;
  if (!c->timing) return WGA_OK;
  const uint32_t n = c->ev_n < (uint32_t)wga_ctx::kTimingRing ? c->ev_n : (uint32_t)wga_ctx::kTimingRing;
  for (uint32_t k = 0; k < n; k++) {
    float ms = 0.0f;
    RT_CHECK(rt_event_elapsed_ms(c->ev[2 * k], c->ev[2 * k + 1], &ms));
    *ms_sum += (double)ms;
  }
  *launches = n;
  c->ev_n = 0;
  return WGA_OK;
}

int wga_scatter_bytes(wga_ctx* c, uint32_t n, const uint8_t* d_src, const uint64_t* d_src_off,
                      uint8_t* d_dst, const uint64_t* d_dst_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (n == 0) return WGA_OK;
  if (!d_src || !d_src_off || !d_dst || !d_dst_off) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  WGA_LAUNCH(k_scatter_bytes, (n + 3u) / 4u, WGA_BLOCK, c->stream, n, d_src, (const u64*)d_src_off,
             d_dst, (const u64*)d_dst_off);
  LAUNCH_CHECK();
  return WGA_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* K3 / K5 / K6 launchers                                                                      */
/* ------------------------------------------------------------------------------------------ */
int wga_maf_pair_stat(wga_ctx* c, uint32_t n, const uint8_t* d_rows, const uint64_t* d_t_off,
                      const uint64_t* d_q_off, const uint64_t* d_cols,
                      const uint8_t* d_strand_neg, wga_cigar_counts* d_counts,
                      uint64_t* d_run_cnt, uint64_t* d_runs, const uint64_t* d_run_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (n == 0) return WGA_OK;
  if (!d_rows || !d_t_off || !d_q_off || !d_cols || !d_strand_neg || !d_counts)
    return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  if (d_runs && !d_run_off) return fail(WGA_E_INVALID_ARG, "d_run_off null", nullptr);
  return maf_walk_call<false>(c, n, d_rows, (const u64*)d_t_off, (const u64*)d_q_off, (const u64*)d_cols, d_strand_neg, d_counts,
                              (u64*)d_run_cnt, (u64*)d_runs, (const u64*)d_run_off);
}

int wga_maf_call_runs(wga_ctx* c, uint32_t n, const uint8_t* d_rows, const uint64_t* d_t_off,
                      const uint64_t* d_q_off, const uint64_t* d_cols, uint64_t* d_run_cnt,
                      uint64_t* d_runs, const uint64_t* d_run_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (n == 0) return WGA_OK;
  if (!d_rows || !d_t_off || !d_q_off || !d_cols) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  if (d_runs && !d_run_off) return fail(WGA_E_INVALID_ARG, "d_run_off null", nullptr);
  return maf_walk_call<true>(c, n, d_rows, (const u64*)d_t_off, (const u64*)d_q_off, (const u64*)d_cols, (const u8*)nullptr,
                             (wga_cigar_counts*)nullptr, (u64*)d_run_cnt, (u64*)d_runs, (const u64*)d_run_off);
}

int wga_cigar_tokenise(wga_ctx* c, uint32_t n, const uint8_t* d_text, const uint64_t* d_text_off,
                       uint64_t* d_op_cnt, wga_tok_err* d_err, uint32_t* d_ops,
                       const uint64_t* d_op_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (n == 0) return WGA_OK;
  if (!d_text || !d_text_off) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  if (d_ops && !d_op_off) return fail(WGA_E_INVALID_ARG, "d_op_off null", nullptr);
  static_assert(sizeof(wga_tok_err) == sizeof(wga_tok_err_dev), "wga_tok_err layout");
  WGA_LAUNCH(k_cigar_tokenise, (n + 3u) / 4u, WGA_BLOCK, c->stream, n, d_text, (const u64*)d_text_off,
             (const u64*)d_text_off + 1, (u64*)d_op_cnt, (wga_tok_err_dev*)d_err, d_ops, (const u64*)d_op_off);
  LAUNCH_CHECK();
  return WGA_OK;
}

int wga_cigar_tokenise_spans(wga_ctx* c, uint32_t n, const uint8_t* d_text, const uint64_t* d_beg,
                             const uint64_t* d_end, uint64_t* d_op_cnt, wga_tok_err* d_err, uint32_t* d_ops,
                             const uint64_t* d_op_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (n == 0) return WGA_OK;
  if (!d_text || !d_beg || !d_end) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  if (d_ops && !d_op_off) return fail(WGA_E_INVALID_ARG, "d_op_off null", nullptr);
  WGA_LAUNCH(k_cigar_tokenise, (n + 3u) / 4u, WGA_BLOCK, c->stream, n, d_text, (const u64*)d_beg,
             (const u64*)d_end, (u64*)d_op_cnt, (wga_tok_err_dev*)d_err, d_ops, (const u64*)d_op_off);
  LAUNCH_CHECK();
  return WGA_OK;
}

int wga_paf_split(wga_ctx* c, const uint8_t* d_text, uint64_t n_bytes, uint64_t* n_lines, wga_paf_line* d_lines,
                  uint64_t cap_lines) {
  static_assert(sizeof(wga_paf_line) == sizeof(wga_paf_line_dev), "wga_paf_line layout");
  return split_lines<0>(c, d_text, n_bytes, n_lines, (void*)d_lines, cap_lines);
}

int wga_maf_split(wga_ctx* c, const uint8_t* d_text, uint64_t n_bytes, uint64_t* n_lines, wga_maf_line* d_lines,
                  uint64_t cap_lines) {
  static_assert(sizeof(wga_maf_line) == sizeof(wga_maf_line_dev), "wga_maf_line layout");
  return split_lines<1>(c, d_text, n_bytes, n_lines, (void*)d_lines, cap_lines);
}

int wga_fasta_pool(wga_ctx* c, const uint8_t* d_text, uint64_t n_bytes, uint64_t* n_contigs, uint64_t* pool_bytes,
                   uint8_t* d_pool, wga_fa_contig* d_contigs) {
  static_assert(sizeof(wga_fa_contig) == sizeof(wga_fa_contig_dev), "wga_fa_contig layout");
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!n_contigs || !pool_bytes) return fail(WGA_E_INVALID_ARG, "null count", nullptr);
  if (n_bytes && !d_text) return fail(WGA_E_INVALID_ARG, "d_text null", nullptr);
  if (n_bytes == 0) {
    *n_contigs = *pool_bytes = 0;
    return WGA_OK;
  }
  const u64 nb64 = (n_bytes + 4095u) / 4096u;
  if (nb64 > 0x7FFFFFFFull) return fail(WGA_E_INVALID_ARG, "text too large for one call", nullptr);
  const u32 nb = (u32)nb64;
  /* scratch: block counts | their exclusive scan (+ total) | scan partials | (count call only) the contig table */
  void* ws;
  const size_t head = ((size_t)nb * 2 + 2 + (size_t)nb / 1024 + 4) * sizeof(u64);
  if ((rc = ctx_scratch(c, head, &ws))) return rc;
  u64* blk = (u64*)ws;
  u64* blk_off = blk + nb;
  u64* partial = blk_off + nb + 1;
  ScanPlain sp;
  sp.in = blk;
  WGA_LAUNCH(k_fa_headers<false>, nb, WGA_BLOCK, c->stream, d_text, (u64)n_bytes, blk, (const u64*)nullptr,
             (wga_fa_contig_dev*)nullptr);
  LAUNCH_CHECK();
  if ((rc = run_scan_ws(c, sp, nb, blk_off, partial))) return rc;
  u64 nh = 0;
  RT_CHECK(rt_d2h(&nh, blk_off + nb, sizeof nh, c->stream));
  wga_fa_contig_dev* contigs = (wga_fa_contig_dev*)d_contigs;
  if (!d_pool) { /* the count call keeps its own contig table in the scratch arena */
    const size_t need = head + 64 + (size_t)nh * sizeof(wga_fa_contig_dev);
    if (c->scratch_cap < need) { /* regrowing frees the arena: start again with room for the table */
      if ((rc = ctx_scratch(c, need, &ws))) return rc;
      blk = (u64*)ws;
      blk_off = blk + nb;
      partial = blk_off + nb + 1;
      sp.in = blk;
      WGA_LAUNCH(k_fa_headers<false>, nb, WGA_BLOCK, c->stream, d_text, (u64)n_bytes, blk, (const u64*)nullptr,
                 (wga_fa_contig_dev*)nullptr);
      LAUNCH_CHECK();
      if ((rc = run_scan_ws(c, sp, nb, blk_off, partial))) return rc;
    }
    contigs = (wga_fa_contig_dev*)((char*)ws + ((head + 63) & ~(size_t)63));
  } else if (nh && !d_contigs) {
    return fail(WGA_E_INVALID_ARG, "d_contigs null", nullptr);
  }
  if (nh) {
    WGA_LAUNCH(k_fa_headers<true>, nb, WGA_BLOCK, c->stream, d_text, (u64)n_bytes, blk, (const u64*)blk_off, contigs);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_fa_header_ends, (u32)((nh + 255) / 256), WGA_BLOCK, c->stream, d_text, (u64)n_bytes, nh, contigs);
    LAUNCH_CHECK();
  }
  WGA_LAUNCH(k_fa_bases<false>, nb, WGA_BLOCK, c->stream, d_text, (u64)n_bytes, nh, contigs, blk, (const u64*)nullptr,
             (u8*)nullptr);
  LAUNCH_CHECK();
  if ((rc = run_scan_ws(c, sp, nb, blk_off, partial))) return rc;
  u64 total = 0;
  RT_CHECK(rt_d2h(&total, blk_off + nb, sizeof total, c->stream));
  *n_contigs = nh;
  *pool_bytes = total;
  if (!d_pool) return WGA_OK;
  WGA_LAUNCH(k_fa_bases<true>, nb, WGA_BLOCK, c->stream, d_text, (u64)n_bytes, nh, contigs, blk, (const u64*)blk_off, d_pool);
  LAUNCH_CHECK();
  if (nh) {
    WGA_LAUNCH(k_fa_finish, (u32)((nh + 255) / 256), WGA_BLOCK, c->stream, (u64)n_bytes, nh, total, contigs);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_fa_lengths, (u32)((nh + 255) / 256), WGA_BLOCK, c->stream, nh, total, contigs);
    LAUNCH_CHECK();
  }
  return WGA_OK;
}

/* K18: bytes in HBM -> BGZF members (wga_k18_bgzf_deflate.h) */
static const uint8_t k_bgzf_eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                       0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
uint64_t wga_bgzf_bound(uint64_t n_bytes) {
  const uint64_t members = (n_bytes + WGA_BGZF_IN - 1u) / WGA_BGZF_IN;
  return n_bytes + members * (uint64_t)(WGA_BGZF_HDR + 5u + WGA_BGZF_TRAILER) + sizeof k_bgzf_eof;
}
int wga_bgzf_compress(wga_ctx* c, const uint8_t* d_in, uint64_t n_bytes, uint8_t* d_out, uint64_t out_cap,
                      uint64_t* out_bytes, int eof_marker) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!out_bytes) return fail(WGA_E_INVALID_ARG, "out_bytes null", nullptr);
  if (n_bytes && !d_in) return fail(WGA_E_INVALID_ARG, "d_in null", nullptr);
  const u64 nb64 = (n_bytes + WGA_BGZF_IN - 1u) / WGA_BGZF_IN;
  if (nb64 > 0x7FFFFFFFull) return fail(WGA_E_INVALID_ARG, "more than 2^31 members in one call", nullptr);
  const u32 nb = (u32)nb64;
  const u64 tail = eof_marker ? sizeof k_bgzf_eof : 0u;
  u64 total = 0;
  if (nb) {
    /* scratch: member sizes | their exclusive scan (+ total) | scan partials | crc + kind per member | code lengths */
    void* ws;
    const size_t words = (size_t)nb * 2 + 2 + (size_t)nb / 1024 + 4;
    const size_t head = words * sizeof(u64);
    const size_t need = head + (size_t)nb * sizeof(wga_bgzf_member) + (size_t)nb * WGA_BGZF_LENS;
    if ((rc = ctx_scratch(c, need, &ws))) return rc;
    u64* sizes = (u64*)ws;
    u64* offs = sizes + nb;
    u64* partial = offs + nb + 1;
    wga_bgzf_member* members = (wga_bgzf_member*)((char*)ws + head);
    u8* lens = (u8*)(members + nb);
    WGA_LAUNCH(k_bgzf_plan, nb, WGA_BLOCK, c->stream, d_in, (u64)n_bytes, sizes, members, lens);
    LAUNCH_CHECK();
    ScanPlain sp;
    sp.in = sizes;
    if ((rc = run_scan_ws(c, sp, nb, offs, partial))) return rc;
    RT_CHECK(rt_d2h(&total, offs + nb, sizeof total, c->stream));
    *out_bytes = total + tail;
    if (!d_out) return WGA_OK; /* the count call */
    if (total + tail > out_cap) return fail(WGA_E_INVALID_ARG, "output buffer smaller than the compressed stream (wga_bgzf_bound)", nullptr);
    WGA_LAUNCH(k_bgzf_emit, nb, WGA_BLOCK, c->stream, d_in, (u64)n_bytes, (const u64*)offs, (const wga_bgzf_member*)members,
               (const u8*)lens, d_out);
    LAUNCH_CHECK();
  }
  *out_bytes = total + tail;
  if (tail && d_out) {
    if (total + tail > out_cap) return fail(WGA_E_INVALID_ARG, "output buffer smaller than the compressed stream (wga_bgzf_bound)", nullptr);
    RT_CHECK(rt_h2d(d_out + total, k_bgzf_eof, sizeof k_bgzf_eof, c->stream));
  }
  return WGA_OK;
}

int wga_pafcov_format(wga_ctx* c, const uint8_t* d_name, uint32_t name_len, const int32_t* d_cov,
                      uint64_t p0, uint32_t count, uint64_t* d_line_off, uint8_t* d_out) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!d_line_off || (count && !d_cov) || (name_len && !d_name)) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  ScanCovLine f;
  f.cov = (const int*)d_cov;
  f.p0 = p0;
  f.name_len = name_len;
  if (!d_out) return run_scan(c, f, count, (u64*)d_line_off);
  if (count == 0) return WGA_OK;
  WGA_LAUNCH(k_pafcov_format, (count + WGA_BED_LINES - 1u) / WGA_BED_LINES, WGA_BLOCK, c->stream, f, count, d_name,
             (const u64*)d_line_off, d_out);
  LAUNCH_CHECK();
  return WGA_OK;
}

int wga_cigar_chain(wga_ctx* c, const wga_cigar_batch* b, wga_chain_trim_t* d_trim, uint64_t* d_nbytes,
                    wga_rec_diag* d_diag, uint8_t* d_out, const uint64_t* d_out_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if ((rc = check_batch(b))) return rc;
  if (b->n == 0) return WGA_OK;
  static_assert(sizeof(wga_chain_trim_t) == sizeof(wga_chain_trim), "wga_chain_trim layout");
  if (!d_out) {
    if (!d_trim || !d_nbytes || !d_diag) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
    RT_CHECK(rt_memset(d_diag, 0xFF, (size_t)b->n * sizeof(wga_rec_diag), c->stream));
    if (!op_all_pieces(c, b))
      WGA_LAUNCH(k_cigar_chain<false>, (b->n + 3u) / 4u, WGA_BLOCK, c->stream, b->n, b->d_ops,
               (const u64*)b->d_op_off, (wga_chain_trim*)d_trim, (u64*)d_nbytes, d_diag, (u8*)nullptr,
               (const u64*)nullptr, (u64)c->op_long_ops);
  } else {
    if (!d_out_off) return fail(WGA_E_INVALID_ARG, "d_out_off null", nullptr);
    if (!op_all_pieces(c, b))
      WGA_LAUNCH(k_cigar_chain<true>, (b->n + 3u) / 4u, WGA_BLOCK, c->stream, b->n, b->d_ops,
               (const u64*)b->d_op_off, (wga_chain_trim*)nullptr, (u64*)nullptr, (wga_rec_diag*)nullptr,
               d_out, (const u64*)d_out_off, (u64)c->op_long_ops);
  }
  LAUNCH_CHECK();
  /* records beyond op_long_ops: pieces over the whole chip, cut where a line is certain; the count call leaves the pieces'
   * places for the fill call (op_piece_table) */
  wga_ctx::OpTabKey key;
  key.kernel = 10, key.ops = b->d_ops, key.op_off = b->d_op_off, key.n = b->n, key.n_ops = b->n_ops;
  key.long_ops = c->op_long_ops, key.piece_ops = c->op_piece_ops;
  bool hit = false;
  if ((rc = op_piece_table(c, b, sizeof(wga_chain_piece), key, d_out != nullptr, &hit))) return rc;
  const wga_ctx::OpTab& t = c->op_tab;
  if (t.np == 0) return WGA_OK;
  wga_chain_piece* pc = (wga_chain_piece*)t.pieces;
  const u32 grid = t.np < 4u * 2048u ? (t.np + 3u) / 4u : 2048u;
  if (!hit) {
    WGA_LAUNCH((k_cigar_chain_pieces<0>), grid, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off,
               (const u64*)t.piece_off, (const u32*)t.piece_rec, pc,
               d_out ? (wga_chain_trim*)nullptr : (wga_chain_trim*)d_trim, d_out ? (wga_rec_diag*)nullptr : d_diag,
               (u8*)nullptr, (const u64*)nullptr);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_cigar_chain_piece_scan, (b->n + 255u) / 256u, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off,
               (const u64*)t.piece_off, pc, d_out ? (wga_chain_trim*)nullptr : (wga_chain_trim*)d_trim,
               d_out ? (u64*)nullptr : (u64*)d_nbytes, d_out ? (wga_rec_diag*)nullptr : d_diag);
    LAUNCH_CHECK();
    op_tab_keep(c, key);
  }
  if (d_out) {
    WGA_LAUNCH((k_cigar_chain_pieces<1>), grid, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off,
               (const u64*)t.piece_off, (const u32*)t.piece_rec, pc, (wga_chain_trim*)nullptr,
               (wga_rec_diag*)nullptr, d_out, (const u64*)d_out_off);
    LAUNCH_CHECK();
  }
  return WGA_OK;
}

int wga_maf_runs_ops(wga_ctx* c, uint32_t n, uint64_t n_elems, const uint64_t* d_runs, const uint64_t* d_run_off,
                     const uint64_t* d_cols, uint64_t* d_cnt, uint32_t* d_out, const uint64_t* d_out_off) {
  if (n && (!d_cols || (n_elems && !d_runs))) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  MafRunOps f;
  f.s = maf_run_src(d_runs, d_run_off, d_cols);
  return run_elems(c, 1, f, n, n_elems, d_run_off, d_cnt, d_out, d_out_off);
}

int wga_maf_runs_cigar_text(wga_ctx* c, uint32_t n, uint64_t n_elems, const uint64_t* d_runs,
                            const uint64_t* d_run_off, const uint64_t* d_cols, uint64_t* d_cnt, uint8_t* d_out,
                            const uint64_t* d_out_off) {
  if (n && (!d_cols || (n_elems && !d_runs))) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  MafRunText f;
  f.s = maf_run_src(d_runs, d_run_off, d_cols);
  return run_elems(c, 2, f, n, n_elems, d_run_off, d_cnt, d_out, d_out_off);
}

int wga_chain_lines_ops(wga_ctx* c, uint32_t n, uint64_t n_elems, const uint64_t* d_lines,
                        const uint64_t* d_line_off, uint64_t* d_cnt, uint32_t* d_out, const uint64_t* d_out_off) {
  if (n && n_elems && !d_lines) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  ChainLineOps f;
  f.s.lines = (const u64*)d_lines;
  return run_elems(c, 3, f, n, n_elems, d_line_off, d_cnt, d_out, d_out_off);
}

int wga_chain_lines_cigar_text(wga_ctx* c, uint32_t n, uint64_t n_elems, const uint64_t* d_lines,
                               const uint64_t* d_line_off, uint64_t* d_cnt, uint8_t* d_out,
                               const uint64_t* d_out_off) {
  if (n && n_elems && !d_lines) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  ChainLineText f;
  f.s.lines = (const u64*)d_lines;
  return run_elems(c, 4, f, n, n_elems, d_line_off, d_cnt, d_out, d_out_off);
}

int wga_cigar_dotplot(wga_ctx* c, const wga_cigar_batch* b, uint64_t cutoff, const uint64_t* d_t_start,
                      const uint64_t* d_q_start, uint64_t* d_seg_cnt, uint64_t* d_segs,
                      const uint64_t* d_seg_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if ((rc = check_batch(b))) return rc;
  if (b->n == 0) return WGA_OK;
  if (!d_t_start || !d_q_start) return fail(WGA_E_INVALID_ARG, "start arrays null", nullptr);
  if (!d_segs) {
    if (!d_seg_cnt) return fail(WGA_E_INVALID_ARG, "d_seg_cnt null", nullptr);
    if (!op_all_pieces(c, b))
      WGA_LAUNCH(k_dotplot_segments<false>, (b->n + 3u) / 4u, WGA_BLOCK, c->stream, b->n, b->d_ops,
               (const u64*)b->d_op_off, b->d_strand_neg, (u64)cutoff, (const u64*)d_t_start,
               (const u64*)d_q_start, (u64*)d_seg_cnt, (u64*)nullptr, (const u64*)nullptr, (u64)c->op_long_ops);
  } else {
    if (!d_seg_off) return fail(WGA_E_INVALID_ARG, "d_seg_off null", nullptr);
    if (!op_all_pieces(c, b))
      WGA_LAUNCH(k_dotplot_segments<true>, (b->n + 3u) / 4u, WGA_BLOCK, c->stream, b->n, b->d_ops,
               (const u64*)b->d_op_off, b->d_strand_neg, (u64)cutoff, (const u64*)d_t_start,
               (const u64*)d_q_start, (u64*)nullptr, (u64*)d_segs, (const u64*)d_seg_off, (u64)c->op_long_ops);
  }
  LAUNCH_CHECK();
  /* records beyond op_long_ops: pieces over the whole chip (as in wga_paf_call_events) */
  wga_ctx::OpTabKey key;
  key.kernel = 12, key.ops = b->d_ops, key.op_off = b->d_op_off, key.n = b->n, key.n_ops = b->n_ops;
  key.x0 = b->d_strand_neg, key.x1 = d_t_start, key.x2 = d_q_start, key.p0 = cutoff;
  key.long_ops = c->op_long_ops, key.piece_ops = c->op_piece_ops;
  bool hit = false;
  if ((rc = op_piece_table(c, b, sizeof(wga_dot_piece), key, d_segs != nullptr, &hit))) return rc;
  const wga_ctx::OpTab& t = c->op_tab;
  if (t.np == 0) return WGA_OK;
  wga_dot_piece* pc = (wga_dot_piece*)t.pieces;
  const u32 grid = t.np < 4u * 2048u ? (t.np + 3u) / 4u : 2048u;
  if (!hit) {
    WGA_LAUNCH((k_dotplot_pieces<0>), grid, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off, b->d_strand_neg,
               (u64)cutoff, (const u64*)d_t_start, (const u64*)d_q_start, (const u64*)t.piece_off, (const u32*)t.piece_rec, pc, (u64*)nullptr, (const u64*)nullptr);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_dotplot_piece_scan, (b->n + 255u) / 256u, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off,
               b->d_strand_neg, (u64)cutoff, (const u64*)d_t_start, (const u64*)d_q_start, (const u64*)t.piece_off, pc,
               d_segs ? (u64*)nullptr : (u64*)d_seg_cnt);
    LAUNCH_CHECK();
    op_tab_keep(c, key);
  }
  if (d_segs) {
    WGA_LAUNCH((k_dotplot_pieces<1>), grid, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off, b->d_strand_neg,
               (u64)cutoff, (const u64*)d_t_start, (const u64*)d_q_start, (const u64*)t.piece_off, (const u32*)t.piece_rec, pc, (u64*)d_segs, (const u64*)d_seg_off);
    LAUNCH_CHECK();
  }
  return WGA_OK;
}

int wga_counts_total(wga_ctx* c, uint32_t n, const wga_cigar_counts* d_counts, uint64_t* d_totals) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (!d_totals || (n && !d_counts)) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  static_assert(sizeof(wga_cigar_counts) == 88, "wga_cigar_counts is 11 u64");
  RT_CHECK(rt_memset(d_totals, 0, 88, c->stream));
  if (n == 0) return WGA_OK;
  u32 grid = (u32)(((u64)n * 11ull + 253ull * 8ull - 1ull) / (253ull * 8ull)); /* ~8 values per thread */
  if (grid > 2048u) grid = 2048u;
  WGA_LAUNCH(k_counts_total, grid, WGA_BLOCK, c->stream, n, (const u64*)d_counts, (u64*)d_totals);
  LAUNCH_CHECK();
  return WGA_OK;
}

int wga_paf_call_events(wga_ctx* c, const wga_cigar_batch* b, uint64_t svlen, int snp,
                        uint64_t* d_ev_cnt, uint64_t* d_ev, const uint64_t* d_ev_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if ((rc = check_batch(b))) return rc;
  if (b->n == 0) return WGA_OK;
  if (d_ev && !d_ev_off) return fail(WGA_E_INVALID_ARG, "d_ev_off null", nullptr);
  if (!op_all_pieces(c, b))
    WGA_LAUNCH(k_paf_call_events, (b->n + 3u) / 4u, WGA_BLOCK, c->stream, b->n, b->d_ops,
             (const u64*)b->d_op_off, (u64)svlen, (u32)(snp != 0), (u64*)d_ev_cnt, (u64*)d_ev,
             (const u64*)d_ev_off, (u64)c->op_long_ops);
  LAUNCH_CHECK();
  /* records beyond op_long_ops: pieces over the whole chip; the count call walks the pieces for their sums and leaves their
   * start states for the fill call (op_piece_table), which walks them again and writes */
  wga_ctx::OpTabKey key;
  key.kernel = 7, key.ops = b->d_ops, key.op_off = b->d_op_off, key.n = b->n, key.n_ops = b->n_ops;
  key.p0 = svlen, key.p1 = snp != 0, key.long_ops = c->op_long_ops, key.piece_ops = c->op_piece_ops;
  bool hit = false;
  if ((rc = op_piece_table(c, b, sizeof(wga_call_piece), key, d_ev != nullptr, &hit))) return rc;
  const wga_ctx::OpTab& t = c->op_tab;
  if (t.np == 0) return WGA_OK;
  wga_call_piece* pc = (wga_call_piece*)t.pieces;
  const u32 grid = t.np < 4u * 2048u ? (t.np + 3u) / 4u : 2048u;
  if (!hit) {
    WGA_LAUNCH((k_paf_call_pieces<0>), grid, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off, (u64)svlen,
               (u32)(snp != 0), (const u64*)t.piece_off, (const u32*)t.piece_rec, pc, (u64*)nullptr,
               (const u64*)nullptr);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_paf_call_piece_scan, (b->n + 255u) / 256u, WGA_BLOCK, c->stream, b->n, (const u64*)t.piece_off, pc,
               d_ev ? (u64*)nullptr : (u64*)d_ev_cnt);
    LAUNCH_CHECK();
    op_tab_keep(c, key);
  }
  if (d_ev) {
    WGA_LAUNCH((k_paf_call_pieces<1>), grid, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off, (u64)svlen,
               (u32)(snp != 0), (const u64*)t.piece_off, (const u32*)t.piece_rec, pc, (u64*)d_ev,
               (const u64*)d_ev_off);
    LAUNCH_CHECK();
  }
  return WGA_OK;
}

int wga_bgzf_inflate(wga_ctx* c, const uint8_t* d_in, uint64_t in_bytes, uint32_t n_blocks, const wga_bgzf_block* d_blocks,
                     uint8_t* d_out, uint32_t* d_status) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (n_blocks == 0) return WGA_OK;
  static_assert(sizeof(wga_bgzf_block) == sizeof(wga_bgzf_block_dev) && sizeof(wga_bgzf_block) == 24, "wga_bgzf_block layout");
  if (!d_in || !d_blocks || !d_out || !d_status) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  WGA_LAUNCH(k_bgzf_inflate, (n_blocks + 3u) / 4u, WGA_BLOCK, c->stream, d_in, (u64)in_bytes, n_blocks,
             (const wga_bgzf_block_dev*)d_blocks, d_out, (u32*)d_status);
  LAUNCH_CHECK();
  return WGA_OK;
}

int wga_paf_call_vcf(wga_ctx* c, const wga_cigar_batch* b, uint64_t svlen, const uint64_t* d_ev, const uint64_t* d_ev_off,
                     const wga_vcf_rec* d_recs, const uint8_t* d_names, const uint8_t* d_t_pool, const uint8_t* d_q_pool,
                     uint64_t* d_nbytes, wga_vcf_err* d_err, uint8_t* d_out, const uint64_t* d_out_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if ((rc = check_batch(b))) return rc;
  if (b->n == 0) return WGA_OK;
  static_assert(sizeof(wga_vcf_rec) == sizeof(wga_vcf_rec_dev) && sizeof(wga_vcf_rec) == 88, "wga_vcf_rec layout");
  static_assert(sizeof(wga_vcf_err) == sizeof(wga_vcf_err_dev) && sizeof(wga_vcf_err) == 16, "wga_vcf_err layout");
  if (!d_ev_off || !d_recs || !d_names || !d_t_pool || !d_q_pool) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  if (!d_out) {
    if (!d_nbytes || !d_err) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
    RT_CHECK(rt_memset(d_err, 0xFF, (size_t)b->n * sizeof(wga_vcf_err), c->stream));
    WGA_LAUNCH(k_paf_call_vcf<false>, (b->n + 3u) / 4u, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off,
               b->d_strand_neg, (u64)svlen, (const u64*)d_ev, (const u64*)d_ev_off, (const wga_vcf_rec_dev*)d_recs, d_names,
               d_t_pool, d_q_pool, (u64*)d_nbytes, (wga_vcf_err_dev*)d_err, (u8*)nullptr, (const u64*)nullptr);
  } else {
    if (!d_out_off) return fail(WGA_E_INVALID_ARG, "d_out_off null", nullptr);
    WGA_LAUNCH(k_paf_call_vcf<true>, (b->n + 3u) / 4u, WGA_BLOCK, c->stream, b->n, b->d_ops, (const u64*)b->d_op_off,
               b->d_strand_neg, (u64)svlen, (const u64*)d_ev, (const u64*)d_ev_off, (const wga_vcf_rec_dev*)d_recs, d_names,
               d_t_pool, d_q_pool, (u64*)nullptr, (wga_vcf_err_dev*)nullptr, d_out, (const u64*)d_out_off);
  }
  LAUNCH_CHECK();
  return WGA_OK;
}

int wga_maf_call_vcf(wga_ctx* c, uint32_t n, const uint8_t* d_rows, const uint64_t* d_t_off, const uint64_t* d_q_off,
                     const uint64_t* d_cols, const uint64_t* d_runs, const uint64_t* d_run_off, const wga_maf_vcf_rec* d_recs,
                     const uint8_t* d_names, int snp, int inv, uint64_t svlen, uint64_t chunk_size, uint64_t* d_nbytes,
                     wga_vcf_err* d_err, uint8_t* d_out, const uint64_t* d_out_off) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (n == 0) return WGA_OK;
  static_assert(sizeof(wga_maf_vcf_rec) == sizeof(wga_maf_vcf_rec_dev) && sizeof(wga_maf_vcf_rec) == 56, "wga_maf_vcf_rec layout");
  if (!d_rows || !d_t_off || !d_q_off || !d_cols || !d_runs || !d_run_off || !d_recs || !d_names)
    return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  if (chunk_size == 0) return fail(WGA_E_INVALID_ARG, "chunk_size must be positive", nullptr);
  if (!d_out) {
    if (!d_nbytes || !d_err) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
    WGA_LAUNCH(k_maf_call_vcf<false>, (n + 3u) / 4u, WGA_BLOCK, c->stream, n, d_rows, (const u64*)d_t_off, (const u64*)d_q_off,
               (const u64*)d_cols, (const u64*)d_runs, (const u64*)d_run_off, (const wga_maf_vcf_rec_dev*)d_recs, d_names,
               (u32)(snp != 0), (u32)(inv != 0), (u64)svlen, (u64)chunk_size, (u64*)d_nbytes, (wga_vcf_err_dev*)d_err, (u8*)nullptr,
               (const u64*)nullptr);
  } else {
    if (!d_out_off) return fail(WGA_E_INVALID_ARG, "d_out_off null", nullptr);
    WGA_LAUNCH(k_maf_call_vcf<true>, (n + 3u) / 4u, WGA_BLOCK, c->stream, n, d_rows, (const u64*)d_t_off, (const u64*)d_q_off,
               (const u64*)d_cols, (const u64*)d_runs, (const u64*)d_run_off, (const wga_maf_vcf_rec_dev*)d_recs, d_names,
               (u32)(snp != 0), (u32)(inv != 0), (u64)svlen, (u64)chunk_size, (u64*)nullptr, (wga_vcf_err_dev*)nullptr, d_out,
               (const u64*)d_out_off);
  }
  LAUNCH_CHECK();
  return WGA_OK;
}

/* The targets' counter ranges [first, one past last] in ascending order, ranges of no counters left out: what the replay's
 * marks -> counts part walks (k_cov_windows<true>).  Ranges that overlap are refused. */
static int cov_ranges(wga_ctx* c, u32 n_targets, const u64* d_cov_off, const u64* d_cov_len, std::vector<u64>& lo_hi,
                      u32* n_rng, u64* n_cov) {
  std::vector<u64> h((size_t)n_targets * 2);
  if (n_targets) {
    RT_CHECK(rt_d2h(h.data(), d_cov_off, (size_t)n_targets * 8, c->stream));
    RT_CHECK(rt_d2h(h.data() + n_targets, d_cov_len, (size_t)n_targets * 8, c->stream));
  }
  std::vector<std::pair<u64, u64>> r;
  r.reserve(n_targets);
  for (u32 t = 0; t < n_targets; t++) {
    const u64 lo = h[t], len = h[(size_t)n_targets + t];
    if (lo + len < lo) return fail(WGA_E_INVALID_ARG, "pafcov: a target's range wraps", nullptr);
    if (len) r.emplace_back(lo, lo + len);
  }
  std::sort(r.begin(), r.end());
  u64 top = 0;
  for (size_t i = 0; i < r.size(); i++) {
    if (r[i].first < top) return fail(WGA_E_INVALID_ARG, "pafcov: target ranges overlap", nullptr);
    top = r[i].second;
  }
  lo_hi.resize(r.size() * 2);
  for (size_t i = 0; i < r.size(); i++) {
    lo_hi[i] = r[i].first;
    lo_hi[r.size() + i] = r[i].second;
  }
  *n_rng = (u32)r.size();
  *n_cov = top;
  return WGA_OK;
}

/* The order in which the marks -> counts replay takes the windows (k_cov_windows<true>): by depth inside their range's chain
 * of windows, chains side by side.  A window whose first counter lies strictly inside a range needs what the window in front
 * hands on and stands one deeper than it; every other window starts a chain.  Kept in the context for the ranges it was made
 * for (a caller's targets do not change between calls). */
static int cov_window_order(wga_ctx* c, const std::vector<u64>& lo_hi, u32 n_rng, u64 nw) {
  std::vector<u64> key(lo_hi);
  key.push_back(nw);
  if (c->cov_order && key == c->cov_order_key) return WGA_OK;
  std::vector<u32> depth((size_t)nw), order((size_t)nw);
  std::vector<u32> cnt;
  u32 t = 0;
  for (u64 w = 0; w < nw; w++) {
    const u64 w0 = w << WGA_COV_WIN_SHIFT;
    while (t < n_rng && lo_hi[(size_t)n_rng + t] <= w0) t++; /* ranges that end at or in front of w0 */
    const bool inside = t < n_rng && lo_hi[t] < w0;          /* lo < w0 < hi */
    const u32 d = (inside && w) ? depth[(size_t)w - 1] + 1u : 0u;
    depth[(size_t)w] = d;
    if (d >= cnt.size()) cnt.resize((size_t)d + 1, 0u);
    cnt[d]++;
  }
  u32 run = 0;
  for (u32& x : cnt) {
    const u32 k = x;
    x = run;
    run += k;
  }
  for (u64 w = 0; w < nw; w++) order[cnt[depth[(size_t)w]]++] = (u32)w;
  if (c->cov_order_cap < nw) {
    if (c->cov_order) RT_CHECK(rt_free(c->cov_order));
    c->cov_order = nullptr;
    c->cov_order_cap = 0;
    RT_CHECK(rt_malloc(&c->cov_order, (size_t)nw * 4));
    c->cov_order_cap = nw;
  }
  c->cov_order_key.clear();
  RT_CHECK(rt_h2d(c->cov_order, order.data(), (size_t)nw * 4, c->stream));
  RT_CHECK(rt_sync(c->stream)); /* `order` is a host buffer of this call */
  c->cov_order_key.swap(key);
  return WGA_OK;
}

/* accumulate (b != nullptr) and / or finalize (n_targets counter ranges): one replay over the windows does both */
static int pafcov_run(wga_ctx* c, const wga_cigar_batch* b, const uint32_t* d_target_id, const uint64_t* d_t_start,
                      const uint64_t* d_cov_off, const uint64_t* d_cov_len, int32_t* d_cov, uint64_t total_cov, bool final,
                      uint32_t n_targets) {
  std::vector<u64> lo_hi;
  u32 n_rng = 0;
  u64 n_cov = total_cov;
  if (final) {
    u64 top = 0;
    int rc = cov_ranges(c, n_targets, (const u64*)d_cov_off, (const u64*)d_cov_len, lo_hi, &n_rng, &top);
    if (rc) return rc;
    if (b && top > total_cov) return fail(WGA_E_INVALID_ARG, "pafcov: a target's range ends behind total_cov", nullptr);
    if (!b) n_cov = top;
  }
  const bool has_ops = b && b->n != 0 && b->n_ops != 0 && total_cov != 0;
  if (!has_ops && (!final || n_rng == 0)) return WGA_OK;
  const u64 nt = has_ops ? ((u64)b->n_ops + WGA_COV_TILE - 1) / WGA_COV_TILE : 0; /* K5 cuts the ops into tiles of its own size */
  const u64 nw = (n_cov >> WGA_COV_WIN_SHIFT) + 1;
  if (nw > 0x7FFFFFFFull) return fail(WGA_E_INVALID_ARG, "coverage arrays too large for one call", nullptr);
  /* One pass lists every (tile, record segment, window) piece (see wga_k5_pafcov.h): tile sums by look-back, the pieces into the
   * tile's own slots and counted under their windows; a scan of the window counts, and the pieces are taken to their windows.
   * Pieces beyond a tile's slots go to one of WGA_COV_LISTS list regions, as large as the last call needed them (+ 25 %): a call
   * that overflows one is run again. */
  void* ws;
  const size_t b_tail = (size_t)nt * 8, b_lcnt = (size_t)WGA_COV_LISTS * 8, b_wcnt = (((size_t)nw * 4) + 15) & ~(size_t)15;
  const size_t b_woff = (((size_t)nw + 1) * 8 + ((size_t)(nw + 1023) / 1024 + 2) * 8 + 15) & ~(size_t)15;
  const size_t b_tcnt = ((size_t)nt * 4 + 15) & ~(size_t)15;
  const size_t b_tinfo = (size_t)nt * sizeof(wga_cov_tile), b_rpos = has_ops ? (size_t)b->n * sizeof(wga_cov_rec) : 0;
  const size_t b_state = final ? (size_t)nw * 8 : 0, b_rng = (size_t)n_rng * 16;
  int rc;
  if ((rc = ctx_scratch(c, b_tail + b_lcnt + b_wcnt + b_woff + b_tcnt + b_tinfo + b_rpos + b_state + b_rng + 64, &ws))) return rc;
  u64* tile_tail = (u64*)ws;
  u64* list_cnt = tile_tail + nt;
  u32* win_cnt = (u32*)(list_cnt + WGA_COV_LISTS);
  u64* win_off = (u64*)((char*)win_cnt + b_wcnt);
  u32* tile_cnt = (u32*)((char*)win_off + b_woff);
  wga_cov_tile* tile_info = (wga_cov_tile*)((char*)tile_cnt + b_tcnt);
  wga_cov_rec* rec_pos = (wga_cov_rec*)((char*)tile_info + b_tinfo);
  u64* win_state = (u64*)((char*)rec_pos + b_rpos);
  u64* rng_lo = (u64*)((char*)win_state + b_state);
  u64* rng_hi = rng_lo + n_rng;
  if (final) {
    if ((rc = cov_window_order(c, lo_hi, n_rng, nw))) return rc;
    RT_CHECK(rt_memset(win_state, 0, b_state, c->stream));
    if (n_rng) RT_CHECK(rt_h2d(rng_lo, lo_hi.data(), b_rng, c->stream));
    RT_CHECK(rt_sync(c->stream)); /* lo_hi is a host buffer of this call */
  }
  u64 n_pieces = 0;
  if (has_ops) {
    const u32 grid = (u32)((nt + WGA_K5_LIST_BW - 1) / WGA_K5_LIST_BW);
    if (c->cov_tile_list_cap < nt) {
      if (c->cov_tile_list) RT_CHECK(rt_free(c->cov_tile_list));
      c->cov_tile_list = nullptr;
      c->cov_tile_list_cap = 0;
      RT_CHECK(rt_malloc(&c->cov_tile_list, (size_t)nt * WGA_COV_TILE_CAP * sizeof(wga_cov_piece)));
      c->cov_tile_list_cap = nt;
    }
    /* what a tile's wave needs of its first two records, in one load: every record's place in the coverage index space, then the
     * record of every tile's first op with its own and its successor's data */
    WGA_LAUNCH(k_cov_rec_pos, (b->n + WGA_BLOCK - 1) / WGA_BLOCK, WGA_BLOCK, c->stream, b->n, d_target_id, (const u64*)d_t_start,
               (const u64*)d_cov_off, (const u64*)d_cov_len, rec_pos);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_cov_tile_info, (b->n + 255u) / 256u, WGA_BLOCK, c->stream, (const u64*)b->d_op_off, b->n, (u64)b->n_ops,
               (const wga_cov_rec*)rec_pos, tile_info);
    LAUNCH_CHECK();
    std::vector<u64> h_cnt(WGA_COV_LISTS);
    u64 n_over = 0;
    for (int attempt = 0;; attempt++) {
      RT_CHECK(rt_memset(ws, 0, b_tail + b_lcnt + b_wcnt, c->stream));
      WGA_LAUNCH(k_cov_list_pieces, grid, 64u * WGA_K5_LIST_BW, c->stream, b->d_ops, (const u64*)b->d_op_off, (u64)b->n_ops,
                 (const wga_cov_tile*)tile_info, (const wga_cov_rec*)rec_pos, tile_tail, win_cnt,
                 (wga_cov_piece*)c->cov_tile_list, tile_cnt, list_cnt, (wga_cov_piece*)c->cov_list, (u64)c->cov_list_rcap,
                 (u32)c->cov_spin_limit);
      LAUNCH_CHECK();
      RT_CHECK(rt_d2h(h_cnt.data(), list_cnt, b_lcnt, c->stream));
      u64 most = 0;
      n_over = 0;
      for (u64 v : h_cnt) {
        n_over += v;
        if (v > most) most = v;
      }
      if (most <= c->cov_list_rcap) break;
      if (attempt) return fail(WGA_E_HIP, "pafcov: the list regions overflow a second time", nullptr);
      if (c->cov_list) RT_CHECK(rt_free(c->cov_list));
      c->cov_list = nullptr;
      c->cov_list_rcap = 0;
      const u64 rcap = most + most / 4 + 16;
      RT_CHECK(rt_malloc(&c->cov_list, (size_t)rcap * WGA_COV_LISTS * sizeof(wga_cov_piece)));
      c->cov_list_rcap = rcap;
    }
    {
      /* run_scan uses the context scratch itself: give it its own small buffer behind win_off */
      ScanU32 f;
      f.in = win_cnt;
      u32 nb = ((u32)nw + 1023u) / 1024u;
      u64* partial = win_off + nw + 1;
      if (nb) {
        WGA_LAUNCH(k_scan_partials<ScanU32>, nb, WGA_BLOCK, c->stream, f, (u32)nw, partial);
        LAUNCH_CHECK();
      }
      WGA_LAUNCH(k_scan_top, 1, WGA_BLOCK, c->stream, partial, nb, win_off + nw);
      LAUNCH_CHECK();
      if (nb) {
        WGA_LAUNCH(k_scan_final<ScanU32>, nb, WGA_BLOCK, c->stream, f, (u32)nw, (const u64*)partial, win_off);
        LAUNCH_CHECK();
      }
    }
    RT_CHECK(rt_d2h(&n_pieces, win_off + nw, sizeof(u64), c->stream));
    if (n_pieces) {
      if (c->cov_pieces_cap < n_pieces) {
        if (c->cov_pieces) RT_CHECK(rt_free(c->cov_pieces));
        c->cov_pieces = nullptr;
        c->cov_pieces_cap = 0;
        RT_CHECK(rt_malloc(&c->cov_pieces, (size_t)(n_pieces + n_pieces / 4) * sizeof(wga_cov_desc)));
        c->cov_pieces_cap = n_pieces + n_pieces / 4;
      }
      RT_CHECK(rt_memset(win_cnt, 0, (size_t)nw * 4, c->stream)); /* now the windows' fill counters */
      WGA_LAUNCH(k_cov_place_tiles, (u32)((nt * WGA_COV_TILE_CAP + WGA_BLOCK - 1) / WGA_BLOCK), WGA_BLOCK, c->stream, (u64)nt,
                 (u64)b->n_ops, (const u32*)tile_cnt, (const wga_cov_piece*)c->cov_tile_list, win_cnt, (const u64*)win_off,
                 (wga_cov_desc*)c->cov_pieces);
      LAUNCH_CHECK();
      if (n_over) {
        dim3 pgrid((u32)((c->cov_list_rcap + WGA_BLOCK - 1) / WGA_BLOCK), WGA_COV_LISTS, 1);
        WGA_LAUNCH(k_cov_place_pieces, pgrid, WGA_BLOCK, c->stream, (u64)b->n_ops, (const u64*)list_cnt,
                   (const wga_cov_piece*)c->cov_list, (u64)c->cov_list_rcap, win_cnt, (const u64*)win_off, (wga_cov_desc*)c->cov_pieces);
        LAUNCH_CHECK();
      }
    }
  }
  const u32* d_ops = has_ops ? b->d_ops : nullptr;
  const u64 n_ops = has_ops ? (u64)b->n_ops : 0;
  const u64* woff = n_pieces ? (const u64*)win_off : nullptr;
  if (final) {
    WGA_LAUNCH(k_cov_windows<true>, (u32)nw, WGA_COV_BLOCK, c->stream, d_ops, n_ops, (const wga_cov_desc*)c->cov_pieces,
               (const wga_cov_piece*)c->cov_tile_list, (const wga_cov_piece*)c->cov_list, woff,
               (int*)d_cov, (u64)n_cov, (const u64*)rng_lo, (const u64*)rng_hi, n_rng, win_state,
#ifdef WGA_COV_NO_ORDER /* A/B builds: the windows in index order */
               (const u32*)nullptr);
#else
               (const u32*)c->cov_order);
#endif
    LAUNCH_CHECK();
  } else if (n_pieces) {
    WGA_LAUNCH(k_cov_windows<false>, (u32)nw, WGA_COV_BLOCK, c->stream, d_ops, n_ops, (const wga_cov_desc*)c->cov_pieces,
               (const wga_cov_piece*)c->cov_tile_list, (const wga_cov_piece*)c->cov_list, woff,
               (int*)d_cov, (u64)n_cov, (const u64*)nullptr, (const u64*)nullptr, 0u, (u64*)nullptr, (const u32*)nullptr);
    LAUNCH_CHECK();
  }
  return WGA_OK;
}

static int pafcov_args(wga_ctx* c, const wga_cigar_batch* b, const void* d_target_id, const void* d_t_start,
                       const void* d_cov_off, const void* d_cov_len, const void* d_cov) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if ((rc = check_batch(b))) return rc;
  if (b->n != 0 && b->n_ops != 0 && (!d_target_id || !d_t_start || !d_cov_off || !d_cov_len || !d_cov))
    return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  return WGA_OK;
}

int wga_pafcov_accumulate(wga_ctx* c, const wga_cigar_batch* b, const uint32_t* d_target_id,
                          const uint64_t* d_t_start, const uint64_t* d_cov_off,
                          const uint64_t* d_cov_len, int32_t* d_cov, uint64_t total_cov) {
  int rc = pafcov_args(c, b, d_target_id, d_t_start, d_cov_off, d_cov_len, d_cov);
  if (rc) return rc;
  return pafcov_run(c, b, d_target_id, d_t_start, d_cov_off, d_cov_len, d_cov, total_cov, false, 0u);
}

int wga_pafcov_accumulate_final(wga_ctx* c, const wga_cigar_batch* b, const uint32_t* d_target_id,
                                const uint64_t* d_t_start, const uint64_t* d_cov_off, const uint64_t* d_cov_len,
                                uint32_t n_targets, int32_t* d_cov, uint64_t total_cov) {
  int rc = pafcov_args(c, b, d_target_id, d_t_start, d_cov_off, d_cov_len, d_cov);
  if (rc) return rc;
  if (n_targets == 0) return WGA_OK;
  if (!d_cov_off || !d_cov_len || !d_cov) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  return pafcov_run(c, b, d_target_id, d_t_start, d_cov_off, d_cov_len, d_cov, total_cov, true, n_targets);
}

int wga_pafcov_finalize(wga_ctx* c, uint32_t n_targets, const uint64_t* d_cov_off,
                        const uint64_t* d_cov_len, int32_t* d_cov) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if (n_targets == 0) return WGA_OK;
  if (!d_cov_off || !d_cov_len || !d_cov) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  return pafcov_run(c, nullptr, nullptr, nullptr, d_cov_off, d_cov_len, d_cov, 0, true, n_targets);
}

/* The class sums are the count call of pafpseudo's protocol (the host sizes the row segments from them): the tile sums and the
 * record sums stay in the context for wga_pafpseudo_fill on the same batch (keyed by its arrays and counts, dropped when one of
 * them is freed), which then is the fill kernel alone. */
int wga_cigar_class_sums(wga_ctx* c, const wga_cigar_batch* b, wga_class_sums* d_sums) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if ((rc = check_batch(b))) return rc;
  if (b->n == 0) return WGA_OK;
  if (!d_sums) return fail(WGA_E_INVALID_ARG, "d_sums null", nullptr);
  wga_ctx::ClassTab& t = c->class_tab;
  t.valid = false;
  u64 nt = n_tiles(b->n_ops);
  if (nt == 0) {
    RT_CHECK(rt_memset(d_sums, 0, (size_t)b->n * sizeof(wga_class_sums), c->stream));
    return WGA_OK;
  }
  const size_t tile_bytes = ((size_t)nt * sizeof(wga_tile_sum) + 63) & ~(size_t)63;
  const size_t need = tile_bytes + (size_t)b->n * sizeof(wga_class_sums);
  if (t.cap < need) {
    if (t.mem) RT_CHECK(rt_free(t.mem));
    t.mem = nullptr;
    t.cap = 0;
    RT_CHECK(rt_malloc(&t.mem, need + need / 4));
    t.cap = need + need / 4;
  }
  t.tiles = (wga_tile_sum*)t.mem;
  t.rec_sums = (wga_class_sums*)((char*)t.mem + tile_bytes);
  RT_CHECK(rt_memset(t.rec_sums, 0, (size_t)b->n * sizeof(wga_class_sums), c->stream));
  WGA_LAUNCH(k_class_tiles, (u32)((nt + 3) / 4), WGA_BLOCK, c->stream, b->d_ops,
             (const u64*)b->d_op_off, b->n, (u64)b->n_ops, t.tiles, t.rec_sums);
  LAUNCH_CHECK();
  static_assert(sizeof(wga_class_sums) % 8 == 0, "wga_class_sums in 64-bit words");
  const u64 words = (u64)b->n * (sizeof(wga_class_sums) / 8);
  WGA_LAUNCH(k_copy_u64, (u32)((words + WGA_BLOCK - 1) / WGA_BLOCK), WGA_BLOCK, c->stream, words, (const u64*)t.rec_sums,
             (u64*)d_sums);
  LAUNCH_CHECK();
  t.ops = b->d_ops;
  t.op_off = b->d_op_off;
  t.n = b->n;
  t.n_ops = b->n_ops;
  t.valid = true;
  return WGA_OK;
}

int wga_pafpseudo_fill(wga_ctx* c, const wga_cigar_batch* b, int base_mode, const uint8_t* d_q_fa,
                       uint64_t q_fa_bytes, const uint64_t* d_q_src_off,
                       const uint64_t* d_q_src_len, const uint64_t* d_skip, uint8_t* d_out,
                       const uint64_t* d_dst_off, wga_rec_diag* d_diag) {
  int rc = ctx_bind(c);
  if (rc) return rc;
  if ((rc = check_batch(b))) return rc;
  if (b->n == 0) return WGA_OK;
  if (!d_skip || !d_out || !d_dst_off || !d_diag) return fail(WGA_E_INVALID_ARG, "null array", nullptr);
  if (base_mode && (!d_q_fa || !d_q_src_off || !d_q_src_len))
    return fail(WGA_E_INVALID_ARG, "base mode needs the query pool", nullptr);
  RT_CHECK(rt_memset(d_diag, 0xFF, (size_t)b->n * sizeof(wga_rec_diag), c->stream));
  u64 nt = n_tiles(b->n_ops);
  if (nt == 0) return WGA_OK;
  wga_tile_sum* tiles;
  wga_class_sums* rec_sums;
  wga_ctx::ClassTab& t = c->class_tab;
  const bool kept = t.valid && t.ops == (const void*)b->d_ops && t.op_off == (const void*)b->d_op_off && t.n == b->n && t.n_ops == b->n_ops;
  t.valid = false; /* one shot: this fill call consumes what the class-sums call left (the sums stay where they are for this call) */
  if (nt > 0x7FFFFFFFull) return fail(WGA_E_INVALID_ARG, "batch too large for one launch", nullptr);
  /* the streaming row kernel (wga_kernels_k2s.h, MODE 2 / 3) with its pre-pass; what it leaves goes to the block kernel */
  const bool stream = c->pseudo_variant == 3;
  const size_t tile_bytes = kept ? 0 : ((size_t)nt * sizeof(wga_tile_sum) + 255) & ~(size_t)255;
  const size_t sums_bytes = kept ? 0 : ((size_t)b->n * sizeof(wga_class_sums) + 255) & ~(size_t)255;
  const size_t rec_bytes = stream ? ((size_t)b->n * sizeof(wga_rec_desc) + 255) & ~(size_t)255 : 0;
  const size_t desc_bytes = stream ? (size_t)nt * sizeof(wga_tile_desc) : 0;
  const size_t list_bytes = stream ? 256 + 2 * (size_t)nt * sizeof(u32) : 0;
  const size_t flag_bytes = stream ? (((size_t)nt + 255) & ~(size_t)255) : 0;
  void* ws = nullptr;
  if (tile_bytes + sums_bytes + rec_bytes + desc_bytes + list_bytes + flag_bytes)
    if ((rc = ctx_scratch(c, tile_bytes + sums_bytes + rec_bytes + desc_bytes + list_bytes + flag_bytes, &ws))) return rc;
  c->pseudo_counts = nullptr;
  if (kept) {
    tiles = t.tiles; /* what wga_cigar_class_sums left for this batch */
    rec_sums = t.rec_sums;
  } else {
    tiles = (wga_tile_sum*)ws;
    rec_sums = (wga_class_sums*)((char*)ws + tile_bytes);
    RT_CHECK(rt_memset(rec_sums, 0, (size_t)b->n * sizeof(wga_class_sums), c->stream));
    WGA_LAUNCH(k_class_tiles, (u32)((nt + 3) / 4), WGA_BLOCK, c->stream, b->d_ops,
               (const u64*)b->d_op_off, b->n, (u64)b->n_ops, tiles, rec_sums);
    LAUNCH_CHECK();
  }
  PseudoArgs a;
  a.ops = b->d_ops;
  a.op_off = (const u64*)b->d_op_off;
  a.strand_neg = b->d_strand_neg;
  a.n = b->n;
  a.n_ops = b->n_ops;
  a.tiles = tiles;
  a.rec_sums = rec_sums;
  a.base_mode = base_mode;
  a.q_fa = d_q_fa;
  a.q_fa_bytes = q_fa_bytes;
  a.q_src_off = (const u64*)d_q_src_off;
  a.q_src_len = (const u64*)d_q_src_len;
  a.skip = (const u64*)d_skip;
  a.out = d_out;
  a.dst_off = (const u64*)d_dst_off;
  a.diag = d_diag;
  a.tile_count = nullptr;
  a.tile_list = nullptr;
  if (stream) {
    char* const base = (char*)ws + tile_bytes + sums_bytes;
    wga_rec_desc* const recs = (wga_rec_desc*)base;
    wga_tile_desc* const tdesc = (wga_tile_desc*)(base + rec_bytes);
    u32* const counts = (u32*)(base + rec_bytes + desc_bytes);
    u32* const list_wide = counts + 64;
    u32* const list_fast = list_wide + nt;
    u8* const tile_flag = (u8*)(base + rec_bytes + desc_bytes + list_bytes);
    RT_CHECK(rt_memset(counts, 0, 256, c->stream));
    RT_CHECK(rt_memset(tile_flag, 0, flag_bytes, c->stream));
    WGA_LAUNCH(k_pseudo_rec_desc, (b->n + 255u) / 256u, WGA_BLOCK, c->stream, b->n, (const wga_class_sums*)rec_sums, b->d_strand_neg,
               base_mode ? (const u64*)d_q_src_off : nullptr, base_mode ? (const u64*)d_q_src_len : nullptr, (const u64*)d_skip,
               (const u64*)d_dst_off, (const u64*)b->d_op_off, (u64)q_fa_bytes, recs, tile_flag);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_tile_base, (u32)((nt + 255) / 256), WGA_BLOCK, c->stream, (const u64*)b->d_op_off, (u64)b->n_ops,
               (const wga_tile_sum*)tiles, (const wga_rec_desc*)recs, tdesc, 1);
    LAUNCH_CHECK();
    WGA_LAUNCH(k_stream_mark_tile, (u32)((nt + 255) / 256), WGA_BLOCK, c->stream, tdesc, (u64)nt, (const u8*)tile_flag, 0, counts,
               list_fast, list_wide);
    LAUNCH_CHECK();
    c->pseudo_counts = counts;
    ExpandArgs e;
    memset(&e, 0, sizeof(e));
    e.ops = b->d_ops;
    e.op_off = (const u64*)b->d_op_off;
    e.n_ops = b->n_ops;
    e.tdesc = tdesc;
    e.recs = recs;
    e.q_fa = base_mode ? d_q_fa : nullptr;
    e.q_fa_bytes = base_mode ? q_fa_bytes : 0;
    e.out = d_out;
    e.diag = d_diag;
    e.n_rec = b->n;
    e.job_tiles = job_tiles_for(c->expand_job_tiles, nt);
    const u64 jobs = (nt + e.job_tiles - 1) / e.job_tiles;
    if (base_mode)
      WGA_LAUNCH(k_pafpseudo_stream, (u32)((jobs + 1) / 2), 128u, c->stream, e);
    else
      WGA_LAUNCH(k_pafpseudo_stream_sym, (u32)((jobs + 1) / 2), 128u, c->stream, e);
    LAUNCH_CHECK();
    const u32 side_grid = nt < 256 ? (u32)nt : 256u;
    a.tile_count = counts; /* tiles of records that are not clean or lie at a pool's edge, tiles beyond 2^24 bases */
    a.tile_list = list_fast;
    if (base_mode)
      WGA_LAUNCH(k_pafpseudo_fill_list<true>, side_grid, WGA_BLOCK, c->stream, a);
    else
      WGA_LAUNCH(k_pafpseudo_fill_list<false>, side_grid, WGA_BLOCK, c->stream, a);
    LAUNCH_CHECK();
    a.tile_count = counts + 1; /* ... beyond 2^31: the block kernel decides on its op-serial walk itself */
    a.tile_list = list_wide;
    if (base_mode)
      WGA_LAUNCH(k_pafpseudo_fill_list<true>, side_grid, WGA_BLOCK, c->stream, a);
    else
      WGA_LAUNCH(k_pafpseudo_fill_list<false>, side_grid, WGA_BLOCK, c->stream, a);
    LAUNCH_CHECK();
    return WGA_OK;
  }
  if (base_mode)
    WGA_LAUNCH(k_pafpseudo_fill<true>, (u32)nt, WGA_BLOCK, c->stream, a);
  else
    WGA_LAUNCH(k_pafpseudo_fill<false>, (u32)nt, WGA_BLOCK, c->stream, a);
  LAUNCH_CHECK();
  return WGA_OK;
}

} /* extern "C" */
