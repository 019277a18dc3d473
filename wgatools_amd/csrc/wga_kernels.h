/*
 * wga_kernels.h — hand-written HIP kernels for gfx950 (CDNA4, wave64) of the wgatools CIGAR hot
 * path.  Integer / byte work, HBM-bound: no MFMA anywhere.  Included by wga_capi.cpp.
 *
 * Work decomposition (all kernels): the packed op stream of a whole batch is cut into
 * *globally aligned tiles* of WGA_TILE ops, independent of record boundaries, so the load is
 * balanced whatever the record-length skew (1-op records and 2 Mop records in one launch).  A
 * tile finds the record of its first op with a 64-ary ballot search over op_off, then walks the
 * record segments it intersects.
 *
 *   K1 k_cigar_stat      one wave per tile; 16 B/lane coalesced op loads; per-segment wave
 *                        reduction; writes per-record counts (atomics only for records that span
 *                        tiles) and an 80-byte tile summary (class sums of the tile and of its
 *                        last segment) that lets any later kernel place a tile inside a long
 *                        record by summing summaries instead of rescanning ops.
 *   K2 k_paf2maf_expand  one 256-thread block per tile; block scan of the tile's ops into LDS
 *                        (column prefix + compacted per-row gap lists); then every thread owns
 *                        16-byte aligned output chunks, binary-searches the gap list in LDS and
 *                        assembles the chunk from <=16-byte source windows (funnel-shifted
 *                        dword loads; reverse-complement fused for '-' strand) — coalesced
 *                        16 B/lane stores, no read-modify-write.
 *
 * The same source also compiles under tests/emu/simt_emu.h (WGA_EMU) for CPU-side logic tests.
 */
#ifndef WGA_KERNELS_H
#define WGA_KERNELS_H

#include "../../include/wga_hip.h"
#include "wga_rt.h"

typedef unsigned int u32;
typedef unsigned long long u64;
typedef long long i64;
typedef unsigned char u8;

#define WGA_TILE 1024u
#define WGA_BLOCK 256u
#define WGA_FAST_COL_LIMIT 0x7FFFFFFFull /* tile-relative columns kept in u32 on the fast path */

/* op classes */
#define CLS_MX 0u
#define CLS_I 1u
#define CLS_D 2u
#define CLS_S 3u
#define CLS_O 4u /* N H P OTHER: consume neither row in paf2maf terms; "move" for pafcov */

/* class of a packed op code (4 bits) — 3 bits per code packed into a 64-bit constant */
__device__ __forceinline__ u32 op_class(u32 code) {
  /* code: 0 M,1 I,2 D,3 N,4 S,5 H,6 P,7 =,8 X,9 Icont,10 Dcont,11.. other */
  const u64 lut = (u64)CLS_MX | ((u64)CLS_I << 3) | ((u64)CLS_D << 6) | ((u64)CLS_O << 9) |
                  ((u64)CLS_S << 12) | ((u64)CLS_O << 15) | ((u64)CLS_O << 18) |
                  ((u64)CLS_MX << 21) | ((u64)CLS_MX << 24) | ((u64)CLS_I << 27) |
                  ((u64)CLS_D << 30) | ((u64)CLS_O << 33) | ((u64)CLS_O << 36) |
                  ((u64)CLS_O << 39) | ((u64)CLS_O << 42) | ((u64)CLS_O << 45);
  return (u32)(lut >> (code * 3)) & 7u;
}

/* tile summary: class sums over the whole tile and over its last record segment */
struct wga_tile_sum {
  u64 tot[5];
  u64 tail[5];
};

/* a 16-byte vector that only promises dword alignment (global_load_dwordx4 needs no more) */
typedef u32 u32x4_a4 __attribute__((vector_size(16), aligned(4)));
typedef u32 u32x4_a16 __attribute__((vector_size(16), aligned(16)));

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
  for (int m = 32; m >= 1; m >>= 1) {
    u32 o = __shfl_xor(v, m);
    v = o < v ? o : v;
  }
  return v;
}

/* Last r in [0, n] with off[r] <= x (off is non-decreasing, off[0] == 0).  Wave-uniform call:
 * a 64-ary search, one coalesced probe + ballot per level. */
__device__ __forceinline__ u32 wga_find_rec(const u64* off, u32 n, u64 x) {
  u32 lane = threadIdx.x & 63u;
  u64 lo = 0, hi = (u64)n + 1;
  while (hi - lo > 1) {
    u64 span = hi - lo;
    u64 step = (span + 63) >> 6;
    u64 p = lo + (u64)(lane + 1) * step;
    int pred = (p < hi) && (off[p] <= x);
    u64 m = __ballot(pred);
    u64 k = (u64)__popcll(m);
    u64 nhi = lo + (k + 1) * step;
    lo = lo + k * step;
    hi = nhi < hi ? nhi : hi;
  }
  return (u32)lo;
}

/* ============================================================================================ */
/* K1: PAF stat walk — parse_paf_to_cigar (cigar.rs:629-707) over packed ops                    */
/* ============================================================================================ */
__global__ __launch_bounds__(256) void k_cigar_stat(const u32* __restrict__ ops,
                                                    const u64* __restrict__ op_off,
                                                    const u8* __restrict__ strand_neg, u32 n,
                                                    u64 n_ops, wga_cigar_counts* counts,
                                                    wga_rec_diag* diag, wga_tile_sum* tiles) {
  const u32 lane = threadIdx.x & 63u;
  const u32 wave = threadIdx.x >> 6;
  const u64 g = (u64)blockIdx.x * 4 + wave;
  const u64 tile_start = g * WGA_TILE;
  if (tile_start >= n_ops) return; /* wave-uniform; this kernel has no block barrier */
  const u64 tile_end = tile_start + WGA_TILE < n_ops ? tile_start + WGA_TILE : n_ops;
  const u32 nt = (u32)(tile_end - tile_start);

  /* 16 ops per lane: op (j*64+lane)*4+e — each of the 4 loads is a fully coalesced 1 KiB */
  u32 w[16];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    u32 base = ((u32)j * 64u + lane) * 4u;
    if (base + 3 < nt) {
      u32x4_a16 v = *(const u32x4_a16*)(ops + tile_start + base);
      w[4 * j + 0] = v[0];
      w[4 * j + 1] = v[1];
      w[4 * j + 2] = v[2];
      w[4 * j + 3] = v[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++)
        w[4 * j + e] = (base + e < nt) ? ops[tile_start + base + e] : 0u; /* 0M: neutral */
    }
  }

  u32 r = wga_find_rec(op_off, n, tile_start);
  u64 cur = tile_start;
  u64 tot[5] = {0, 0, 0, 0, 0}, tail[5] = {0, 0, 0, 0, 0};
  while (cur < tile_end) {
    u64 re = op_off[r + 1];
    while (re <= cur) { /* skip empty records */
      r++;
      re = op_off[r + 1];
    }
    const u64 rs = op_off[r];
    const u64 seg_end = re < tile_end ? re : tile_end;
    const u32 a = (u32)(cur - tile_start), b = (u32)(seg_end - tile_start);

    /* per-lane partials: 16 ops * (2^28-1) < 2^32, so u32 is exact */
    u32 s[5] = {0, 0, 0, 0, 0};
    u32 s_match = 0; /* M,= only (X is the rest of CLS_MX) */
    u32 ev = 0;      /* ins events | del events << 16 */
    u32 bad = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        u32 idx = ((u32)j * 64u + lane) * 4u + (u32)e;
        u32 op = w[4 * j + e];
        u32 code = op & 15u, len = op >> 4;
        bool in = idx >= a && idx < b;
        u32 cls = op_class(code);
        u32 l = in ? len : 0u;
        s[0] += cls == CLS_MX ? l : 0u;
        s[1] += cls == CLS_I ? l : 0u;
        s[2] += cls == CLS_D ? l : 0u;
        s[3] += cls == CLS_S ? l : 0u;
        s[4] += cls == CLS_O ? l : 0u;
        s_match += (code == WGA_OP_M || code == WGA_OP_EQ) ? l : 0u;
        ev += (in && code == WGA_OP_I) ? 1u : 0u;
        ev += (in && code == WGA_OP_D) ? 0x10000u : 0u;
        bool isbad = in && (cls == CLS_S || cls == CLS_O);
        bad = (isbad && idx < bad) ? idx : bad;
      }
    }
    u64 S[5];
#pragma unroll
    for (int c = 0; c < 5; c++) S[c] = wave_sum_u64((u64)s[c]);
    const u64 Smatch = wave_sum_u64((u64)s_match);
    const u32 EV = wave_sum_u32(ev);
    const u32 BAD = wave_min_u32(bad);

    if (lane == 0) {
      const bool neg = strand_neg[r] != 0;
      const bool whole = rs >= tile_start && re <= tile_end;
      const u64 match = Smatch, mism = S[0] - Smatch;
      const u64 iev = EV & 0xFFFFu, dev = EV >> 16;
      wga_cigar_counts* c = counts + r;
      if (whole) { /* the record lives in this tile only: plain stores */
        c->match = match;
        c->mismatch = mism;
        c->ins_ev = neg ? 0 : iev;
        c->ins_bp = neg ? 0 : S[1];
        c->del_ev = neg ? 0 : dev;
        c->del_bp = neg ? 0 : S[2];
        c->inv_ins_ev = neg ? iev : 0;
        c->inv_ins_bp = neg ? S[1] : 0;
        c->inv_del_ev = neg ? dev : 0;
        c->inv_del_bp = neg ? S[2] : 0;
        c->inv_ev = neg ? 1 : 0;
      } else { /* record spans tiles: counts were zeroed by the launcher */
        u64* f = (u64*)c;
        if (match) atomicAdd(f + 0, match);
        if (mism) atomicAdd(f + 1, mism);
        const int o = neg ? 4 : 0; /* ins_ev.. -> inv_ins_ev.. (cigar.rs:667-684) */
        if (iev) atomicAdd(f + 2 + o, iev);
        if (S[1]) atomicAdd(f + 3 + o, S[1]);
        if (dev) atomicAdd(f + 4 + o, dev);
        if (S[2]) atomicAdd(f + 5 + o, S[2]);
        if (neg && rs >= tile_start) atomicAdd(f + 10, (u64)1); /* inv_event = 1, once */
      }
      if (BAD != 0xFFFFFFFFu) atomicMin((u64*)&diag[r].bad_op_idx, tile_start + BAD - rs);
    }
#pragma unroll
    for (int c = 0; c < 5; c++) {
      tot[c] += S[c];
      tail[c] = S[c];
    }
    cur = seg_end;
    r++;
  }
  if (tiles && lane == 0) {
    wga_tile_sum ts;
#pragma unroll
    for (int c = 0; c < 5; c++) {
      ts.tot[c] = tot[c];
      ts.tail[c] = tail[c];
    }
    tiles[g] = ts;
  }
}

/* ============================================================================================ */
/* block-level exclusive scan helpers (256 threads)                                             */
/* ============================================================================================ */
__device__ __forceinline__ u64 block_excl_scan_u64(u64 v, u64* s_w /*[5]*/, u64* total) {
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  u64 inc = v;
  for (u32 d = 1; d < 64; d <<= 1) {
    u64 t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  __syncthreads(); /* protect s_w reuse */
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  u64 pre = 0, tot = 0;
  for (u32 k = 0; k < 4; k++) {
    u64 x = s_w[k];
    if (k < wave) pre += x;
    tot += x;
  }
  *total = tot;
  return pre + inc - v;
}

/* ---- generic exclusive scan of n u64 values: 3 kernels, 1024 values per block --------------- */
struct ScanPlain {
  const u64* in;
  __device__ u64 operator()(u32 i) const { return in[i]; }
};

/* value functor of the paf2maf layout: bytes one record occupies in the output text */
struct ScanLayout {
  const wga_cigar_counts* counts;
  const u64* t_src_len;
  const u64* q_src_len;
  const u32* pre_t;
  const u32* pre_q;
  const u32* post;
  __device__ u64 t_row(u32 i) const { /* String::insert_str grows the target by the I bases */
    return t_src_len[i] + counts[i].ins_bp + counts[i].inv_ins_bp;
  }
  __device__ u64 q_row(u32 i) const { return q_src_len[i] + counts[i].del_bp + counts[i].inv_del_bp; }
  __device__ u64 operator()(u32 i) const {
    return (u64)(pre_t ? pre_t[i] : 0u) + t_row(i) + (u64)(pre_q ? pre_q[i] : 0u) + q_row(i) +
           (u64)(post ? post[i] : 0u);
  }
};

template <typename F>
__global__ __launch_bounds__(256) void k_scan_partials(F f, u32 n, u64* partial) {
  __shared__ u64 s_w[5];
  u32 base = blockIdx.x * 1024u + threadIdx.x * 4u;
  u64 v = 0;
  for (u32 e = 0; e < 4; e++)
    if (base + e < n) v += f(base + e);
  u64 tot;
  (void)block_excl_scan_u64(v, s_w, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

/* single block: exclusive scan of nb partials in place, grand total to *total_out */
__global__ __launch_bounds__(256) void k_scan_top(u64* partial, u32 nb, u64* total_out) {
  __shared__ u64 s_w[5];
  u64 carry = 0;
  for (u32 base = 0; base < nb; base += 256u) {
    u32 i = base + threadIdx.x;
    u64 v = i < nb ? partial[i] : 0;
    u64 tot;
    u64 ex = block_excl_scan_u64(v, s_w, &tot);
    if (i < nb) partial[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

template <typename F>
__global__ __launch_bounds__(256) void k_scan_final(F f, u32 n, const u64* partial, u64* out) {
  __shared__ u64 s_w[5];
  u32 base = blockIdx.x * 1024u + threadIdx.x * 4u;
  u64 x[4];
  u64 v = 0;
  for (u32 e = 0; e < 4; e++) {
    x[e] = (base + e < n) ? f(base + e) : 0;
    v += x[e];
  }
  u64 tot;
  u64 ex = block_excl_scan_u64(v, s_w, &tot) + partial[blockIdx.x];
  for (u32 e = 0; e < 4; e++) {
    if (base + e < n) out[base + e] = ex;
    ex += x[e];
  }
}

/* row offsets from record offsets (converter.rs:237-262 + maf.rs:566-581 geometry) */
__global__ __launch_bounds__(256) void k_layout_rows(ScanLayout f, u32 n, const u64* rec_off,
                                                     u64* t_row_off, u64* q_row_off) {
  u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  u64 t = rec_off[i] + (f.pre_t ? f.pre_t[i] : 0u);
  t_row_off[i] = t;
  q_row_off[i] = t + f.t_row(i) + (f.pre_q ? f.pre_q[i] : 0u);
}

/* ============================================================================================ */
/* K2: paf2maf gap insertion                                                                    */
/* ============================================================================================ */
struct RowSrc {
  const u8* fa;  /* sequence pool */
  u64 fa_bytes;  /* pool size (window loads are bounds-checked against it) */
  u64 src_off;   /* start of this record's slice in the pool */
  u64 src_len;   /* slice length as fetched */
  bool rc;       /* read reversed + complemented (utils.rs:83-101) */
};

/* 0x80 in every byte of y that is zero (exact: no cross-byte carries) */
__device__ __forceinline__ u32 zero_bytes(u32 y) {
  return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

/* complement 4 packed bases; *valid gets 0x80 per byte that is one of ACGTNacgtn */
__device__ __forceinline__ u32 comp4(u32 x, u32* valid) {
  u32 low = x | 0x20202020u;
  u32 isn = zero_bytes(low ^ 0x6E6E6E6Eu);
  u32 v = isn | zero_bytes(low ^ 0x61616161u) | zero_bytes(low ^ 0x63636363u) |
          zero_bytes(low ^ 0x67676767u) | zero_bytes(low ^ 0x74747474u);
  *valid = v;
  u32 b1 = (x >> 1) & 0x01010101u;         /* bit 1: set for C/G, clear for A/T */
  u32 xm = 0x15151515u - b1 * 0x11u;       /* A<->T: ^0x15, C<->G: ^0x04 */
  u32 nmask = (isn >> 7) * 0xFFu;          /* N stays N */
  return x ^ (xm & ~nmask);
}

__device__ __forceinline__ u32 bswap32(u32 x) {
  return (x >> 24) | ((x >> 8) & 0xFF00u) | ((x << 8) & 0xFF0000u) | (x << 24);
}
__device__ __forceinline__ u32 alignbyte(u32 hi, u32 lo, u32 sh) {
  return (u32)(((((u64)hi) << 32) | (u64)lo) >> (8u * sh));
}
/* bytes [lo,hi) ∩ [0,4) of a dword as a 0xFF mask */
__device__ __forceinline__ u32 bytemask(int lo, int hi) {
  lo = lo < 0 ? 0 : lo;
  hi = hi > 4 ? 4 : hi;
  if (hi <= lo) return 0u;
  u32 mh = hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u);
  u32 ml = (1u << (8 * lo)) - 1u;
  return mh & ~ml;
}

/* Load the 16 source bytes that map to output bytes 0..15 of a chunk: byte j <-> slice index
 * S + j (S may be negative / beyond the slice for bytes outside the piece — those are masked by
 * the caller; memory safety comes from the pool bounds check).  For rc the slice is read
 * backwards and complemented; *inv gets 0x80 flags of invalid bases (all 16 bytes). */
__device__ __forceinline__ void load_window(const RowSrc& src, i64 S, int pa, int pb, u32 W[4],
                                            u32 inv[4]) {
  /* pool index of the lowest-addressed byte of the window */
  i64 P = src.rc ? (i64)src.src_off + (i64)src.src_len - 16 - S : (i64)src.src_off + S;
  u64 addr = (u64)src.fa + (u64)P;
  u64 al = addr & ~3ull;
  u32 V[4];
  if (P >= 4 && (u64)P + 24 <= src.fa_bytes) { /* whole dword-aligned 20-byte span in the pool */
    u32x4_a4 v = *(const u32x4_a4*)al;
    u32 v4 = *(const u32*)(al + 16);
    u32 sh = (u32)(addr & 3ull);
    V[0] = alignbyte(v[1], v[0], sh);
    V[1] = alignbyte(v[2], v[1], sh);
    V[2] = alignbyte(v[3], v[2], sh);
    V[3] = alignbyte(v4, v[3], sh);
  } else { /* pool edge: guarded byte loads, only for the bytes of the piece */
    V[0] = V[1] = V[2] = V[3] = 0u;
    for (int j = pa; j < pb; j++) {
      int vj = src.rc ? 15 - j : j; /* position inside the address-ordered window */
      i64 idx = P + vj;
      u32 byte = (idx >= 0 && (u64)idx < src.fa_bytes) ? (u32)src.fa[idx] : 0u;
      V[vj >> 2] |= byte << (8 * (vj & 3));
    }
  }
  if (src.rc) {
    u32 r0 = bswap32(V[3]), r1 = bswap32(V[2]), r2 = bswap32(V[1]), r3 = bswap32(V[0]);
    u32 v0, v1, v2, v3;
    W[0] = comp4(r0, &v0);
    W[1] = comp4(r1, &v1);
    W[2] = comp4(r2, &v2);
    W[3] = comp4(r3, &v3);
    inv[0] = ~v0 & 0x80808080u;
    inv[1] = ~v1 & 0x80808080u;
    inv[2] = ~v2 & 0x80808080u;
    inv[3] = ~v3 & 0x80808080u;
  } else {
    W[0] = V[0];
    W[1] = V[1];
    W[2] = V[2];
    W[3] = V[3];
    inv[0] = inv[1] = inv[2] = inv[3] = 0u;
  }
}

__device__ __forceinline__ void merge16(u32 o[4], const u32 W[4], int pa, int pb) {
  if (pa <= 0 && pb >= 16) {
    o[0] = W[0];
    o[1] = W[1];
    o[2] = W[2];
    o[3] = W[3];
    return;
  }
#pragma unroll
  for (int d = 0; d < 4; d++) {
    u32 m = bytemask(pa - 4 * d, pb - 4 * d);
    o[d] = (o[d] & ~m) | (W[d] & m);
  }
}

/*
 * Emit N (<= 2^31) bytes of one gapped row to dst.  Output byte k is tile-relative column
 * c0 + k.  The row's events inside this range are entries [ga, gb) of a compacted list
 * (G_col = tile-relative start column, G_cum = exclusive prefix of gap bases — an entry's gap
 * length is G_cum[i+1]-G_cum[i], entry gb is readable — and G_adj = exclusive prefix of the
 * source adjustment: gap bases minus skipped source bases, as wrapping u32; for paf2maf rows
 * G_adj == G_cum).  A non-gap column c reads slice index
 *     sbase + (c - c_org) - (adj before c - gcum_a)          (gcum_a = adj at c_org)
 * Threads tid, tid+nthreads, ... own 16-byte *address-aligned* chunks of dst.
 */
__device__ __forceinline__ void emit_row(u8* dst, u32 N, u32 c0, u32 c_org, const u32* G_col,
                                         const u32* G_cum, const u32* G_adj, int ga, int gb,
                                         u32 gcum_a, u64 sbase, const RowSrc& src, u32 tid,
                                         u32 nthreads, u64* bad_base_pos) {
  if (N == 0) return;
  const u64 A = (u64)dst, E = A + N;
  const u64 first = A >> 4, last = (E - 1) >> 4;
  for (u64 ch = first + tid; ch <= last; ch += nthreads) {
    const u64 base_addr = ch << 4;
    const u32 a0 = base_addr < A ? (u32)(A - base_addr) : 0u;
    const u32 b0 = base_addr + 16 > E ? (u32)(E - base_addr) : 16u;
    const u32 cz = c0 + (u32)(base_addr - A); /* column of chunk byte 0 (wraps for the head) */
    u32 c = cz + a0;
    const u32 c_end = cz + b0;
    /* last gap entry in [ga, gb) that starts at or before c */
    int lo = ga, hi = gb;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (G_col[mid] <= c)
        lo = mid + 1;
      else
        hi = mid;
    }
    int i = lo - 1;
    bool in_gap = false;
    u32 gap_end = 0, cum = gcum_a;
    if (i >= ga) {
      u32 gs = G_col[i];
      u32 gl = G_cum[i + 1] - G_cum[i];
      if (c - gs < gl) {
        in_gap = true;
        gap_end = gs + gl;
      }
      cum = G_adj[i + 1];
    }
    u32 o[4] = {0u, 0u, 0u, 0u};
    while (c < c_end) {
      if (in_gap) {
        u32 pe = gap_end < c_end ? gap_end : c_end;
        const u32 dash[4] = {0x2D2D2D2Du, 0x2D2D2D2Du, 0x2D2D2D2Du, 0x2D2D2D2Du};
        merge16(o, dash, (int)(c - cz), (int)(pe - cz));
        c = pe;
        in_gap = false;
      } else {
        u32 next_gs = (i + 1 < gb) ? G_col[i + 1] : 0xFFFFFFFFu;
        u32 pe = next_gs < c_end ? next_gs : c_end;
        if (pe > c) {
          const int pa = (int)(c - cz), pb = (int)(pe - cz);
          const i64 S = (i64)sbase + (i64)(int)(cz - c_org) - (i64)(int)(cum - gcum_a);
          u32 W[4], inv[4];
          load_window(src, S, pa, pb, W, inv);
          if (src.rc) { /* InvalidBase: first offender in reversed order = smallest q' index */
#pragma unroll
            for (int d = 0; d < 4; d++) {
              u32 bad = inv[d] & bytemask(pa - 4 * d, pb - 4 * d);
              if (bad) {
                int j = 4 * d + ((__ffsll((unsigned long long)bad) - 1) >> 3);
                atomicMin(bad_base_pos, (u64)(S + j));
              }
            }
          }
          merge16(o, W, pa, pb);
          c = pe;
        }
        if (c < c_end) { /* c == start of gap i+1 */
          i++;
          u32 gs = G_col[i];
          u32 gl = G_cum[i + 1] - G_cum[i];
          if (gl) {
            in_gap = true;
            gap_end = gs + gl;
          }
          cum = G_adj[i + 1];
        }
      }
    }
    if (a0 == 0u && b0 == 16u) {
      u32x4_a16 v = {o[0], o[1], o[2], o[3]};
      *(u32x4_a16*)base_addr = v;
    } else { /* partial chunk at a row / tile edge: byte stores, never read-modify-write */
      u8* p = (u8*)base_addr;
      for (u32 j = a0; j < b0; j++) {
        u32 d = j >> 2;
        u32 word = d == 0 ? o[0] : d == 1 ? o[1] : d == 2 ? o[2] : o[3];
        p[j] = (u8)(word >> (8u * (j & 3u)));
      }
    }
  }
}

/* tail of a row when the fetched slice is longer than the CIGAR consumes: plain copy */
__device__ __forceinline__ void emit_tail(u8* dst, u64 n, u64 sbase, const RowSrc& src, u32 tid,
                                          u32 nthreads, u64* bad_base_pos) {
  u64 done = 0;
  while (done < n) {
    u64 m = n - done;
    if (m > (1ull << 30)) m = 1ull << 30;
    emit_row(dst + done, (u32)m, 0u, 0u, (const u32*)0, (const u32*)0, (const u32*)0, 0, 0, 0u,
             sbase + done, src, tid, nthreads, bad_base_pos);
    done += m;
  }
}

/* one source byte of a row (slow path / tails): slice index -> byte, with rc + validation */
__device__ __forceinline__ u8 src_byte(const RowSrc& src, u64 sidx, u64* bad_base_pos) {
  if (sidx >= src.src_len) return (u8)'?'; /* only reachable for records flagged as panic */
  u64 raw = src.rc ? src.src_len - 1 - sidx : sidx;
  u64 idx = src.src_off + raw;
  u8 c = idx < src.fa_bytes ? src.fa[idx] : (u8)0;
  if (!src.rc) return c;
  u8 o;
  switch (c) {
    case 'A': o = 'T'; break;
    case 'C': o = 'G'; break;
    case 'G': o = 'C'; break;
    case 'T': o = 'A'; break;
    case 'N': o = 'N'; break;
    case 'a': o = 't'; break;
    case 'c': o = 'g'; break;
    case 'g': o = 'c'; break;
    case 't': o = 'a'; break;
    case 'n': o = 'n'; break;
    default:
      o = c;
      atomicMin(bad_base_pos, sidx);
  }
  return o;
}

struct ExpandArgs {
  const u32* ops;
  const u64* op_off;
  const u8* strand_neg;
  u32 n;
  u64 n_ops;
  const wga_cigar_counts* counts;
  const wga_tile_sum* tiles;
  const u8* t_fa;
  u64 t_fa_bytes;
  const u64* t_src_off;
  const u64* t_src_len;
  const u8* q_fa;
  u64 q_fa_bytes;
  const u64* q_src_off;
  const u64* q_src_len;
  u8* out;
  const u64* t_row_off;
  const u64* q_row_off;
  wga_rec_diag* diag;
  int force_slow;
};

__global__ __launch_bounds__(256) void k_paf2maf_expand(ExpandArgs a) {
  __shared__ u32 s_col[WGA_TILE + 1];    /* tile-relative exclusive column prefix per op      */
  __shared__ u32 s_ev[WGA_TILE + 1];     /* exclusive (#I-class ops | #D-class ops << 16)     */
  __shared__ u32 s_tg_col[WGA_TILE + 1]; /* target-row gaps (I ops): start column             */
  __shared__ u32 s_tg_cum[WGA_TILE + 1]; /*                           gap bases before        */
  __shared__ u32 s_qg_col[WGA_TILE + 1]; /* query-row gaps (D ops)                            */
  __shared__ u32 s_qg_cum[WGA_TILE + 1];
  __shared__ u64 s_w[5];
  __shared__ u64 s_red[4][3];

  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u, wave = tid >> 6;
  const u64 g = blockIdx.x;
  const u64 tile_start = g * WGA_TILE;
  const u64 tile_end = tile_start + WGA_TILE < a.n_ops ? tile_start + WGA_TILE : a.n_ops;
  const u32 nt = (u32)(tile_end - tile_start);

  const wga_tile_sum tsum = a.tiles[g];
  const u64 tile_cols = tsum.tot[CLS_MX] + tsum.tot[CLS_I] + tsum.tot[CLS_D];
  const bool fast = !a.force_slow && tile_cols <= WGA_FAST_COL_LIMIT;

  /* ---- phase A: 4 consecutive ops per thread, block scan into LDS -------------------------- */
  u32 opw[4];
  {
    u32 base = tid * 4u;
    if (base + 3 < nt) {
      u32x4_a16 v = *(const u32x4_a16*)(a.ops + tile_start + base);
      opw[0] = v[0];
      opw[1] = v[1];
      opw[2] = v[2];
      opw[3] = v[3];
    } else {
      for (int e = 0; e < 4; e++) opw[e] = (base + e < nt) ? a.ops[tile_start + base + e] : 0u;
    }
  }
  u32 e_col[4], e_i[4], e_d[4], cls[4];
  if (fast) {
    u32 l[4], sl = 0, si = 0, sd = 0, cnt = 0;
    for (int e = 0; e < 4; e++) {
      u32 code = opw[e] & 15u, len = opw[e] >> 4;
      cls[e] = op_class(code);
      l[e] = (cls[e] <= CLS_D) ? len : 0u;
      sl += l[e];
      si += cls[e] == CLS_I ? len : 0u;
      sd += cls[e] == CLS_D ? len : 0u;
      cnt += cls[e] == CLS_I ? 1u : (cls[e] == CLS_D ? 0x10000u : 0u);
    }
    /* one u64 block scan carries (cols | I bases << 32); a second carries (D bases | counts<<32):
     * every component stays below 2^31 on the fast path, so the packed lanes never carry over */
    u64 totA, totB;
    u64 exA = block_excl_scan_u64((u64)sl | ((u64)si << 32), s_w, &totA);
    u64 exB = block_excl_scan_u64((u64)sd | ((u64)cnt << 32), s_w, &totB);
    u32 x_col = (u32)exA, x_i = (u32)(exA >> 32), x_d = (u32)exB, x_cnt = (u32)(exB >> 32);
    for (int e = 0; e < 4; e++) {
      u32 k = tid * 4u + (u32)e;
      e_col[e] = x_col;
      e_i[e] = x_i;
      e_d[e] = x_d;
      s_col[k] = x_col;
      s_ev[k] = x_cnt;
      if (cls[e] == CLS_I) {
        s_tg_col[x_cnt & 0xFFFFu] = x_col;
        s_tg_cum[x_cnt & 0xFFFFu] = x_i;
        x_i += opw[e] >> 4;
        x_cnt += 1u;
      } else if (cls[e] == CLS_D) {
        s_qg_col[x_cnt >> 16] = x_col;
        s_qg_cum[x_cnt >> 16] = x_d;
        x_d += opw[e] >> 4;
        x_cnt += 0x10000u;
      }
      x_col += l[e];
    }
    if (tid == WGA_BLOCK - 1) { /* sentinels: totals */
      s_col[WGA_TILE] = x_col;
      s_ev[WGA_TILE] = x_cnt;
      s_tg_col[x_cnt & 0xFFFFu] = x_col;
      s_tg_cum[x_cnt & 0xFFFFu] = x_i;
      s_qg_col[x_cnt >> 16] = x_col;
      s_qg_cum[x_cnt >> 16] = x_d;
    }
  }
  __syncthreads();

  /* ---- phase B: walk the record segments of this tile ------------------------------------- */
  u32 r = wga_find_rec(a.op_off, a.n, tile_start);
  u64 cur = tile_start;
  while (cur < tile_end) {
    u64 re = a.op_off[r + 1];
    while (re <= cur) {
      r++;
      re = a.op_off[r + 1];
    }
    const u64 rs = a.op_off[r];
    const u64 seg_end = re < tile_end ? re : tile_end;
    const u32 ka = (u32)(cur - tile_start), kb = (u32)(seg_end - tile_start);

    /* class sums of this record before the tile: summaries of tiles g0..g-1 (tail of g0) */
    u64 b_mx = 0, b_i = 0, b_d = 0;
    if (rs < tile_start) { /* block-uniform: only the first segment can continue a record */
      const u64 g0 = rs / WGA_TILE;
      u64 p_mx = 0, p_i = 0, p_d = 0;
      for (u64 k = g0 + tid; k < g; k += WGA_BLOCK) {
        const wga_tile_sum* t = a.tiles + k;
        const u64* v = (k == g0) ? t->tail : t->tot;
        p_mx += v[CLS_MX];
        p_i += v[CLS_I];
        p_d += v[CLS_D];
      }
      p_mx = wave_sum_u64(p_mx);
      p_i = wave_sum_u64(p_i);
      p_d = wave_sum_u64(p_d);
      __syncthreads();
      if (lane == 0) {
        s_red[wave][0] = p_mx;
        s_red[wave][1] = p_i;
        s_red[wave][2] = p_d;
      }
      __syncthreads();
      for (int w2 = 0; w2 < 4; w2++) {
        b_mx += s_red[w2][0];
        b_i += s_red[w2][1];
        b_d += s_red[w2][2];
      }
    }
    const u64 cb = b_mx + b_i + b_d; /* record-relative column of the segment start */
    const u64 tb = b_mx + b_d;       /* target bases consumed before it              */
    const u64 qb = b_mx + b_i;       /* query bases consumed before it               */

    const wga_cigar_counts cn = a.counts[r];
    const u64 I_total = cn.ins_bp + cn.inv_ins_bp, D_total = cn.del_bp + cn.inv_del_bp;
    const u64 L = cn.match + cn.mismatch + I_total + D_total;
    RowSrc ts, qs;
    ts.fa = a.t_fa;
    ts.fa_bytes = a.t_fa_bytes;
    ts.src_off = a.t_src_off[r];
    ts.src_len = a.t_src_len[r];
    ts.rc = false;
    qs.fa = a.q_fa;
    qs.fa_bytes = a.q_fa_bytes;
    qs.src_off = a.q_src_off[r];
    qs.src_len = a.q_src_len[r];
    qs.rc = a.strand_neg[r] != 0;
    const u64 t_row_len = ts.src_len + I_total, q_row_len = qs.src_len + D_total;
    u8* const t_dst = a.out + a.t_row_off[r];
    u8* const q_dst = a.out + a.q_row_off[r];
    u64* const bad_base = (u64*)&a.diag[r].bad_base_pos;
    u64* const panic_idx = (u64*)&a.diag[r].panic_op_idx;

    if (fast) {
      const u32 col_a = s_col[ka], seg_cols = s_col[kb] - col_a;
      const u32 eva = s_ev[ka], evb = s_ev[kb];
      const int ia = (int)(eva & 0xFFFFu), ib = (int)(evb & 0xFFFFu);
      const int ja = (int)(eva >> 16), jb = (int)(evb >> 16);
      const u32 icum_a = s_tg_cum[ia], dcum_a = s_qg_cum[ja];

      /* String::insert_str panics when the insertion point is beyond the string
       * (cigar.rs:507,513): an I (D) op whose target (query) consumption so far exceeds the
       * fetched slice */
      for (int e = 0; e < 4; e++) {
        u32 k = tid * 4u + (u32)e;
        if (k >= ka && k < kb) {
          if (cls[e] == CLS_I) {
            u64 t_before = tb + (u64)(e_col[e] - col_a) - (u64)(e_i[e] - icum_a);
            if (t_before > ts.src_len) atomicMin(panic_idx, tile_start + k - rs);
          } else if (cls[e] == CLS_D) {
            u64 q_before = qb + (u64)(e_col[e] - col_a) - (u64)(e_d[e] - dcum_a);
            if (q_before > qs.src_len) atomicMin(panic_idx, tile_start + k - rs);
          }
        }
      }

      /* rows end where the slice ends (a CIGAR that consumes more than was fetched) */
      u64 x1t = cb + seg_cols < t_row_len ? cb + seg_cols : t_row_len;
      if (x1t > cb)
        emit_row(t_dst + cb, (u32)(x1t - cb), col_a, col_a, s_tg_col, s_tg_cum, s_tg_cum, ia, ib,
                 icum_a, tb, ts, tid, WGA_BLOCK, bad_base);
      u64 x1q = cb + seg_cols < q_row_len ? cb + seg_cols : q_row_len;
      if (x1q > cb)
        emit_row(q_dst + cb, (u32)(x1q - cb), col_a, col_a, s_qg_col, s_qg_cum, s_qg_cum, ja, jb,
                 dcum_a, qb, qs, tid, WGA_BLOCK, bad_base);
    } else {
      /* u64 fallback for tiles wider than 2^31 columns: ops are walked serially (every thread
       * redundantly), each op's columns are written block-strided, one byte per store */
      u64 x = cb, tp = tb, qp = qb;
      for (u64 k = cur; k < seg_end; k++) {
        const u32 op = a.ops[k];
        const u32 c = op_class(op & 15u);
        const u64 len = op >> 4;
        if (c == CLS_I && tp > ts.src_len && tid == 0) atomicMin(panic_idx, k - rs);
        if (c == CLS_D && qp > qs.src_len && tid == 0) atomicMin(panic_idx, k - rs);
        if (c <= CLS_D) {
          for (u64 j = tid; j < len; j += WGA_BLOCK) {
            if (x + j < t_row_len) t_dst[x + j] = (c == CLS_I) ? (u8)'-' : src_byte(ts, tp + j, bad_base);
            if (x + j < q_row_len) q_dst[x + j] = (c == CLS_D) ? (u8)'-' : src_byte(qs, qp + j, bad_base);
          }
          x += len;
          if (c != CLS_I) tp += len;
          if (c != CLS_D) qp += len;
        }
      }
    }

    /* the record ends in this tile: append what the slices hold beyond the CIGAR */
    if (seg_end == re) {
      if (t_row_len > L) emit_tail(t_dst + L, t_row_len - L, L - I_total, ts, tid, WGA_BLOCK, bad_base);
      if (q_row_len > L) emit_tail(q_dst + L, q_row_len - L, L - D_total, qs, tid, WGA_BLOCK, bad_base);
    }
    cur = seg_end;
    r++;
  }
}

/* ---- copy n variable-length snippets (MAF line text between the rows) ------------------------ */
__global__ __launch_bounds__(256) void k_scatter_bytes(u32 n, const u8* src, const u64* src_off,
                                                       u8* dst, const u64* dst_off) {
  /* one wave per snippet; snippets are tens of bytes */
  const u32 lane = threadIdx.x & 63u;
  const u64 i = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const u64 s0 = src_off[i], s1 = src_off[i + 1], d0 = dst_off[i];
  for (u64 k = s0 + lane; k < s1; k += 64) dst[d0 + (k - s0)] = src[k];
}

#endif /* WGA_KERNELS_H */
