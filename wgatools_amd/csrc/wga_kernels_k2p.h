/*
 * wga_kernels_k2p.h — K2 `k_paf2maf_expand_p`: the paf2maf row kernel, second design.
 *
 * What the counters said about v1 (wga_kernels.h; profiles/r02_k2_counters.txt): 3.4e9 VALU + 2.0e9 SALU
 * wave-instructions per launch keep the vector pipe 70-78 % busy, and only a quarter of them move bytes — phase A,
 * the per-segment set-up and above all the row-job header (descriptor fields picked out of lanes, SGPR spills) are
 * executed by all four waves of a block for every job of every segment, whoever ends up emitting it.  And
 * (scripts/micro/line_split_copy.hip) a 128-byte line whose sixteen-byte granules reach the L2 in different store
 * instructions a memory round trip apart is written back in pieces: 19.3 GB of writes for 14.9 GB of rows.
 *
 * This kernel keeps v1's tile analysis (phase A: compacted per-row gap lists + granule table) and changes the rest:
 *   * PLAN.  Wave 0 alone walks the tile's record segments, checks the insert_str panics, cuts the rows into
 *     pieces (1, 2 or 4 per row segment, cut where the OUTPUT ADDRESS is a multiple of 128) and leaves one
 *     64-byte descriptor per piece in LDS; the other waves wait at the barrier and issue nothing.  After the
 *     barrier every wave fetches the descriptors of ITS pieces (round-robin) with one LDS read and emits them:
 *     the set-up runs once per tile instead of four times, the header once per piece by its owner.
 *   * LINE-COMPLETE STORES.  A step of the emitter covers 64 x WGA_P_U consecutive column granules.  The
 *     granules that touch an event or a row edge are worked out FIRST (one lane each, two source windows under
 *     byte masks) into sixteen-byte result slots in LDS; then the plain granules are loaded, and every lane
 *     stores its granule — plain bytes from its window, or the result slot — in the same instruction: the step's
 *     1-4 KB leave as whole lines.  (More than 64 such granules in one step — indel-dense stretches — overflow
 *     into v1's behaviour: stored by the lane that computed them.)
 *   * Tiles narrower than 65 536 columns — all but pathological ones — keep their gap lists as u16 (22 KB of LDS
 *     per block); wider tiles (u32 lists) and tiles beyond 2^31 columns (v1's op-serial walk) are listed by
 *     k_list_wide_tiles and run in two side launches that are normally empty.
 *   * Rows at a pool edge (window loads would need bounds checks) and slice tails beyond the CIGAR — neither
 *     occurs in a consistent PAF — are written byte by byte (emit_bytes_p).
 */
#ifndef WGA_KERNELS_K2P_H
#define WGA_KERNELS_K2P_H

#include "wga_kernels.h"

typedef unsigned short u16;

#ifndef WGA_P_U
#define WGA_P_U 4 /* column granules per lane and step */
#endif
#define WGA_P_PER (64u * WGA_P_U)
#ifndef WGA_K2P_BLOCKS
#define WGA_K2P_BLOCKS 5 /* 96 VGPRs; 6 and 7 blocks (80 / 72 VGPRs) spill and measure no faster */
#endif
#ifndef WGA_K2P_SPLIT
#define WGA_K2P_SPLIT 6144u /* rows beyond this many bytes are cut in two pieces, beyond four times it in four */
#endif
#define WGA_NARROW_COLS 65536ull /* tiles below this many columns take the u16 instance */
#define WGA_P_PIECES 16u         /* piece descriptors per planning round */
#define WGA_P_PIECE_MAX 10u      /* pieces one record segment can yield: 2 rows x 4 + 2 tails */

template <typename GT>
struct RowDescS {
  u32 c_org;
  const GT* G_col; /* event start column (tile relative)                                     */
  const GT* G_cum; /* gap bases before the event                                             */
  const GT* G_adj; /* source adjustment before the event (paf2maf rows: == G_cum)            */
  int ga, gb;
  u32 gcum_a;
  u64 sbase;       /* slice index of column c_org (invalid-base positions are reported relative to the slice) */
  const u32x4_a16* lowmask;
  const u32* tbl;
  u32 tsh, gsh;
  u16* queue;      /* this wave's queue: WGA_P_PER granule indices of the current step */
  u32x4_a16* res;  /* this wave's result slots: the first 64 queued granules of the step, sixteen bytes each */
};

template <typename GT>
__device__ __forceinline__ int find_entry_s(const RowDescS<GT>& rd, u32 c) {
  const int ga = rd.ga, gb = rd.gb;
  const u32 j = c >> rd.gsh;
  int k = (int)((rd.tbl[j] >> rd.tsh) & WGA_TBL_CNT);
  const int hi = (int)((rd.tbl[j + 1] >> rd.tsh) & WGA_TBL_CNT);
  while (k < hi && (u32)rd.G_col[k] <= c) k++;
  k -= 1;
  return k < ga ? ga - 1 : (k >= gb ? gb - 1 : k);
}

/* window post-processing with the strand as a compile-time constant */
template <bool RC>
__device__ __forceinline__ void win_finish_t(const u32 r[4], u32 W[4], u32 inv[4]) {
  if (RC) {
    W[0] = comp4(bswap32(r[3]), &inv[0]);
    W[1] = comp4(bswap32(r[2]), &inv[1]);
    W[2] = comp4(bswap32(r[1]), &inv[2]);
    W[3] = comp4(bswap32(r[0]), &inv[3]);
  } else {
    W[0] = r[0];
    W[1] = r[1];
    W[2] = r[2];
    W[3] = r[3];
    inv[0] = inv[1] = inv[2] = inv[3] = 0u;
  }
}

/* generic piece walk of one granule from an arbitrary state (any number of pieces); rows whose windows need no
 * bounds checks only (the others take emit_bytes_s) */
template <bool RC, typename GT>
__device__ __forceinline__ void emit_walk_s(u32 o[4], u32 c, u32 c_end, u32 cz, int i, bool in_gap,
                                            u32 gap_end, u32 cum, const RowDescS<GT>& rd,
                                            const RowBufs& rb, u64* bad_base_pos) {
  while (c < c_end) {
    if (in_gap) {
      u32 pe = gap_end < c_end ? gap_end : c_end;
      merge_dash(o, (int)(c - cz), (int)(pe - cz), rd.lowmask);
      c = pe;
      in_gap = false;
    } else {
      u32 next_gs = (i + 1 < rd.gb) ? (u32)rd.G_col[i + 1] : 0xFFFFFFFFu;
      u32 pe = next_gs < c_end ? next_gs : c_end;
      if (pe > c) {
        const int pa = (int)(c - cz), pb = (int)(pe - cz);
        const int off = (int)(cz - rd.c_org) - (int)(cum - rd.gcum_a);
        u32 raw[4], W[4], inv[4];
        buf_load16(rb.lbuf, rowbuf_loff(rb, off), raw);
        win_finish_t<RC>(raw, W, inv);
        if (RC) flag_bad_bases(inv, pa, pb, (i64)rd.sbase + off, rd.lowmask, bad_base_pos);
        merge16(o, W, pa, pb, rd.lowmask);
        c = pe;
      }
      if (c < c_end) { /* c == start of entry i+1 */
        i++;
        u32 gs = rd.G_col[i];
        u32 gl = (u32)rd.G_cum[i + 1] - (u32)rd.G_cum[i];
        if (gl) {
          in_gap = true;
          gap_end = gs + gl;
        }
        cum = rd.G_adj[i + 1];
      }
    }
  }
}

/* A granule that touches an event boundary or a row edge:  [gap0 rest] copy0 | gap1 | copy1  from two source
 * windows under byte masks (the straight-line scheme of complex_chunk in wga_kernels.h); bytes outside
 * [g.a0, g.b0) of the result are don't-care (never stored). */
template <bool RC, typename GT>
__device__ __forceinline__ void complex_granule_s(const ChunkGeom& g, const RowDescS<GT>& rd,
                                                  const RowBufs& rb, u64* bad_base_pos, u32 o[4]) {
  const int ga = rd.ga, gb = rd.gb;
  const u32 c = g.c, c_end = g.c_end, cz = g.cz;
  const int i = find_entry_s(rd, c);
  const bool has0 = i >= ga;
  const int ic = has0 ? i : ga; /* always a readable index */
  const u32 gs0 = rd.G_col[ic], cum0a = rd.G_cum[ic], cum0b = rd.G_cum[ic + 1];
  const u32 adj0b = rd.G_adj[ic + 1];
  const u32 gl0 = cum0b - cum0a;
  const bool in_gap0 = has0 && (c - gs0 < gl0);
  const u32 adj0 = has0 ? adj0b : rd.gcum_a;
  const int n1 = i + 1;
  const u32 gs1 = n1 < gb ? (u32)rd.G_col[n1] : 0xFFFFFFFFu;
  const u32 gl1 = (u32)rd.G_cum[n1 + 1] - (u32)rd.G_cum[n1]; /* two sentinels: readable up to gb + 1 */
  const u32 adj1 = rd.G_adj[n1 + 1];
  const u32 gs2 = n1 + 1 < gb ? (u32)rd.G_col[n1 + 1] : 0xFFFFFFFFu;
  const bool hasB = gs1 < c_end;
  const u32 g0e = gs0 + gl0;
  const u32 a1 = in_gap0 ? (g0e < c_end ? g0e : c_end) : c;       /* copy piece 0 = [a1, b1) */
  const u32 b1 = hasB ? gs1 : c_end;
  const u32 g1e = gs1 + gl1;
  const u32 e1 = hasB ? (g1e < c_end ? g1e : c_end) : c_end;      /* gap 1 = [b1, e1)        */
  const u32 b2 = hasB ? (gs2 < c_end ? gs2 : c_end) : c_end;      /* copy piece 1 = [e1, b2) */
  const int offz = (int)(cz - rd.c_org);
  const int off0 = offz - (int)(adj0 - rd.gcum_a), off1 = offz - (int)(adj1 - rd.gcum_a);
  u32 r0[4], r1[4];
  buf_load16(rb.lbuf, b1 > a1 ? rowbuf_loff(rb, off0) : WGA_BUF_OOB, r0);
  buf_load16(rb.lbuf, b2 > e1 ? rowbuf_loff(rb, off1) : WGA_BUF_OOB, r1);
  const u32x4_a16 La1 = rd.lowmask[a1 - cz], Lb1 = rd.lowmask[b1 - cz], Le1 = rd.lowmask[e1 - cz],
                  Lb2 = rd.lowmask[b2 - cz];
  u32 W0[4], W1[4], inv0[4], inv1[4];
  win_finish_t<RC>(r0, W0, inv0);
  win_finish_t<RC>(r1, W1, inv1);
  u32 bad = 0u;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const u32 m0 = Lb1[d] & ~La1[d], m1 = Lb2[d] & ~Le1[d];
    const u32 md = bfi32(Lb1[d], La1[d], Le1[d]); /* [0, a1) + [b1, e1) */
    o[d] = bfi32(m0, W0[d], bfi32(m1, W1[d], md & 0x2D2D2D2Du));
    bad |= (inv0[d] & m0) | (inv1[d] & m1);
  }
  if (RC && bad) { /* rare: report the first invalid base (utils.rs:97) */
    flag_bad_bases(inv0, (int)(a1 - cz), (int)(b1 - cz), (i64)rd.sbase + off0, rd.lowmask, bad_base_pos);
    flag_bad_bases(inv1, (int)(e1 - cz), (int)(b2 - cz), (i64)rd.sbase + off1, rd.lowmask, bad_base_pos);
  }
  if (b2 < c_end) /* a third event inside 16 columns: rare, generic walk from there */
    emit_walk_s<RC>(o, b2, c_end, cz, n1, false, 0u, adj1, rd, rb, bad_base_pos);
}

/* The same N bytes one at a time, by one wave: rows at a pool edge (their windows would need bounds checks) and
 * what a slice holds beyond its CIGAR (ga == gb: no events).  Neither occurs in a consistent PAF whose
 * sequences sit inside the pool; correctness only. */
template <typename GT>
__device__ __forceinline__ void emit_bytes_p(u8* dst, u32 N, u32 c0, const RowDescS<GT>& rd, const RowSrc& src,
                                             u64* bad_base_pos) {
  const u32 lane = threadIdx.x & 63u;
#pragma clang loop vectorize(disable) unroll(disable)
  for (u32 x = lane; x < N; x += 64u) {
    const u32 c = c0 + x;
    u32 adj = rd.gcum_a;
    bool gap = false;
    if (rd.gb > rd.ga) {
      const int i = find_entry_s(rd, c);
      if (i >= rd.ga) {
        const u32 gs = rd.G_col[i], gl = (u32)rd.G_cum[i + 1] - (u32)rd.G_cum[i];
        gap = c - gs < gl;
        adj = rd.G_adj[i + 1];
      }
    }
    dst[x] = gap ? (u8)'-' : src_byte(src, rd.sbase + (u64)(c - rd.c_org) - (u64)(adj - rd.gcum_a), bad_base_pos);
  }
}


/* bytes [a0, b0) of a granule's sixteen at p: the head / tail granule of a row piece; never read-modify-write */
__device__ __forceinline__ void store_bytes16(u8* p, const u32 o[4], u32 a0, u32 b0) {
#pragma clang loop vectorize(disable) unroll(disable)
  for (u32 b = a0; b < b0; b++) {
    const u32 d = b >> 2;
    const u32 word = d == 0 ? o[0] : d == 1 ? o[1] : d == 2 ? o[2] : o[3];
    p[b] = (u8)(word >> (8u * (b & 3u)));
  }
}

/* One row piece = N bytes at dst whose first byte is tile-relative column c0, emitted by ONE wave.  The row's
 * source windows must not need bounds checks (RowSrc::safe); win_base as rowsrc_prepare leaves it.
 * RC = the row is read reverse-complemented. */
template <bool RC, typename GT>
__device__ __forceinline__ void emit_piece_p(u8* dst, u32 N, u32 c0, const RowDescS<GT>& rd,
                                             const u8* win_base, u64* bad_base_pos) {
  const u32 lane = threadIdx.x & 63u;
  const RowGeom rg = row_geom(dst, N, c0);
  u16* const queue = rd.queue;
  u32x4_a16* const res = rd.res;
  const u32 niter = (rg.nchunks + WGA_P_PER - 1u) / WGA_P_PER;
  const u32 lo_full = rg.head == 0u ? 0u : 1u;
  const u32 n_full = rg.nchunks - lo_full - (rg.last_b0 == 16u ? 0u : 1u); /* may wrap to "none" */
  const bool any_full = rg.nchunks >= lo_full + (rg.last_b0 == 16u ? 0u : 1u) + 1u;
  const int koff = (int)(rd.gcum_a - rd.c_org); /* window offset of a granule = cz + koff - adj */
  RowBufs rb;
  rb.sgn = RC ? 0xFFFFFFFFu : 0u;
  rb.kbias = RC ? 0x80000000u : 64u;
  rb.lbuf = buf_make(win_base - (i64)rb.kbias, 0xFFFFFFF0u);
  rb.sbuf = buf_make(rg.base, rg.nchunks << 4);
#pragma nounroll
  for (u32 it = 0; it < niter; it++) {
    const u32 rel0 = it * WGA_P_PER;
    u32 qn = 0; /* wave-uniform queue length */
    /* ---- 1. classify the step's granules; queue those that touch an event or a row edge ---- */
    u32 loff[WGA_P_U];
    u32 info = 0u; /* per granule one byte: bits 0-5 result slot, 6 = queued, 7 = dashes */
#pragma unroll
    for (int u = 0; u < WGA_P_U; u++) {
      const u32 rel = rel0 + (u32)u * 64u + lane;
      const bool act = rel < rg.nchunks;
      const u32 relc = act ? rel : 0u;
      const u32 cz = (rg.j0 + relc) << 4;
      const u32 jg = cz >> rd.gsh;
      u32 w0 = rd.tbl[jg] >> rd.tsh, w1 = rd.tbl[jg + 1] >> rd.tsh;
      WGA_PIN(w0);
      WGA_PIN(w1);
      u32 adj = rd.G_adj[w0 & WGA_TBL_CNT];
      WGA_PIN(adj);
      const u32 st = w1 & (WGA_TBL_COVER | WGA_TBL_FULL);
      const bool dash = st == (WGA_TBL_COVER | WGA_TBL_FULL);
      /* bitwise, not &&: short-circuit evaluation would come back as exec-mask branches */
      const bool cand = (bool)((int)any_full & (int)(rel - lo_full < n_full) & (int)(((w0 ^ w1) & WGA_TBL_CNT) == 0u) &
                               (int)(st != WGA_TBL_COVER));
      const u32 off = cz + (u32)koff - adj; /* slice index of the granule relative to sbase, >= 0 */
      loff[u] = ((int)cand & (int)!dash) ? rowbuf_loff(rb, (int)off) : WGA_BUF_OOB;
      const bool cx = (bool)((int)act & (int)!cand);
      const u64 m = __ballot(cx);
      const u32 slot = qn + lane_rank(m, lane);
      if (cx) queue[slot] = (u16)((u32)u * 64u + lane);
      /* slots beyond 63 have no result slot: their granule is stored by the lane that works it out (code 0x3F) */
      const u32 code = cx ? (0x40u | (slot < 63u ? slot : 63u)) : ((dash & cand) ? 0x80u : 0u);
      info |= code << (8 * u);
      qn += (u32)__popcll(m);
    }
    WGA_WAVE_SYNC();
    /* ---- 2. the queued granules: the first 63 into result slots, the rest (indel-dense stretches) stored here.
     *         (Issuing the plain granules' loads before or together with this step measured 8-9 % slower.) ---- */
#if defined(WGA_P_ABLATE) && WGA_P_ABLATE == 3
    qn = 0u;
#endif
    const u32 take = qn < 63u ? qn : 63u;
    if (lane < take) {
      u32 o[4];
      complex_granule_s<RC>(chunk_geom(rg, rel0 + queue[lane]), rd, rb, bad_base_pos, o);
      const u32x4_a16 ov = {o[0], o[1], o[2], o[3]};
      res[lane] = ov;
    }
    for (u32 q0 = 63u; q0 < qn; q0 += 64u) {
      if (q0 + lane < qn) {
        const u32 rel = rel0 + queue[q0 + lane];
        const ChunkGeom g = chunk_geom(rg, rel);
        u32 o[4];
        complex_granule_s<RC>(g, rd, rb, bad_base_pos, o);
        const bool whole = g.a0 == 0u && g.b0 == 16u;
        buf_store16(rb.sbuf, whole ? rel << 4 : WGA_BUF_OOB, o);
        if (!whole) store_bytes16(g.p, o, g.a0, g.b0);
      }
    }
    WGA_WAVE_SYNC();
    u32 raw[WGA_P_U][4];
#pragma unroll
    for (int u = 0; u < WGA_P_U; u++) buf_load16(rb.lbuf, loff[u], raw[u]);
    /* ---- 3. every lane stores its granule, plain or from its result slot: the step leaves as whole lines ---- */
    u32 badm = 0u;
#pragma unroll
    for (int u = 0; u < WGA_P_U; u++) {
      const u32 code = (info >> (8 * u)) & 0xFFu;
      const u32 rel = rel0 + (u32)u * 64u + lane;
      u32 o[4], inv[4];
      win_finish_t<RC>(raw[u], o, inv);
      if (RC) badm |= loff[u] != WGA_BUF_OOB ? (inv[0] | inv[1] | inv[2] | inv[3]) : 0u;
      const u32x4_a16 rv = res[code & 0x3Fu]; /* unconditional: no exec-mask branch around the LDS read */
      const bool queued = (code & 0x40u) != 0u, dashes = (code & 0x80u) != 0u;
#pragma unroll
      for (int d = 0; d < 4; d++) o[d] = queued ? rv[d] : (dashes ? 0x2D2D2D2Du : o[d]);
      const bool mine = (bool)((int)(loff[u] != WGA_BUF_OOB) | (int)dashes | (int)(queued & ((code & 0x3Fu) != 0x3Fu)));
      const bool part = (bool)((int)queued & ((int)((rel == 0u) & (rg.head != 0u)) | (int)((rel == rg.nchunks - 1u) & (rg.last_b0 != 16u))));
      buf_store16_stream(rb.sbuf, ((int)mine & (int)!part) ? rel << 4 : WGA_BUF_OOB, o);
      if ((int)mine & (int)part) { /* the piece's head / tail granule */
        const ChunkGeom g = chunk_geom(rg, rel);
        store_bytes16(g.p, o, g.a0, g.b0);
      }
    }
    if (RC && badm != 0u) { /* InvalidBase (utils.rs:97), rare: find the first offender of the plain granules again */
#pragma unroll
      for (int u = 0; u < WGA_P_U; u++) {
        u32 rr[4], o[4], inv[4];
        buf_load16(rb.lbuf, loff[u], rr);
        win_finish_t<RC>(rr, o, inv);
        if (loff[u] != WGA_BUF_OOB && (inv[0] | inv[1] | inv[2] | inv[3]) != 0u) {
          const u32 x = loff[u] - rb.kbias;
          flag_bad_bases(inv, 0, 16, (i64)rd.sbase + (int)((x ^ rb.sgn) - rb.sgn), rd.lowmask, bad_base_pos);
        }
      }
    }
    WGA_WAVE_SYNC(); /* queue and result slots are rewritten by the next step */
  }
}


struct ExpandArgsP {
  const u32* ops;
  const u64* op_off;
  u64 n_ops;
  const wga_tile_desc* tdesc;
  const wga_rec_desc* recs;
  const u8* t_fa;
  u64 t_fa_bytes;
  const u8* q_fa;
  u64 q_fa_bytes;
  u8* out;
  wga_rec_diag* diag;
  int no_table;          /* test knob: 256-column granules (the coarse-table path of very wide tiles) */
  const u32* wide_count; /* u32 instance: number of listed tiles, and the list */
  const u32* wide_list;
};

/* v with lane K's copy replaced by a wave-uniform value */
template <u32 K>
__device__ __forceinline__ u32 lane_put_u32(u32 v, u32 uniform_val, u32 lane) {
#ifdef WGA_EMU
  return lane == K ? uniform_val : v;
#else
  (void)lane;
  const u32 uv = WGA_UNI32(uniform_val);
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(uv), "n"(K));
  return v;
#endif
}

/* piece descriptor: 16 dwords in LDS, written by the planner from wave-uniform values, one dword per lane */
#define PD_DST 0      /* 0-1  address of the piece's first byte                       */
#define PD_N 2        /*      bytes                                                   */
#define PD_C0 3       /*      tile-relative column of the first byte                  */
#define PD_CORG 4     /*      column where the row segment starts (slice index sbase) */
#define PD_GCUM 5     /*      gap bases of this row before the segment                */
#define PD_GAB 6      /*      ga | gb << 16: the segment's entries of the gap list    */
#define PD_FLAGS 7    /*      1 query row, 2 reverse-complemented, 4 safe, 8 tail     */
#define PD_SBASE 8    /* 8-9  slice index of column c_org                             */
#define PD_WIN 10     /* 10-11 RowSrc::win_base                                       */
#define PD_REC 12     /*      record                                                  */
__device__ __forceinline__ void piece_put(u32* slot, u32 lane, u64 dst, u32 N, u32 c0, u32 c_org, u32 gcum_a, u32 gab,
                                          u32 flags, u64 sbase, u64 win, u32 rec) {
  u32 v = 0u;
  v = lane_put_u32<0u>(v, (u32)dst, lane);
  v = lane_put_u32<1u>(v, (u32)(dst >> 32), lane);
  v = lane_put_u32<2u>(v, N, lane);
  v = lane_put_u32<3u>(v, c0, lane);
  v = lane_put_u32<4u>(v, c_org, lane);
  v = lane_put_u32<5u>(v, gcum_a, lane);
  v = lane_put_u32<6u>(v, gab, lane);
  v = lane_put_u32<7u>(v, flags, lane);
  v = lane_put_u32<8u>(v, (u32)sbase, lane);
  v = lane_put_u32<9u>(v, (u32)(sbase >> 32), lane);
  v = lane_put_u32<10u>(v, (u32)win, lane);
  v = lane_put_u32<11u>(v, (u32)(win >> 32), lane);
  v = lane_put_u32<12u>(v, rec, lane);
  if (lane < 16u) slot[lane] = v;
}

/* the tile; GT = u16 (tile_cols < 65 536) or u32 (anything up to WGA_FAST_COL_LIMIT) */
template <typename GT>
__device__ __forceinline__ void expand_tile_p(const ExpandArgsP& a, const u64 g, u32 pre, GT* s_tg_col,
                                              GT* s_tg_cum, GT* s_qg_col, GT* s_qg_cum, u32* s_tbl,
                                              u32* s_bnd0, u32* s_tot, u32* s_zero2, u32* s_w4, GT* s_pcol, u32* s_pcnt,
                                              u32* s_plan, u32* s_piece, const u32x4_a16* s_lowmask, u16* s_queue,
                                              u32x4_a16* s_res) {
  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  const u64 tile_start = g * WGA_TILE;
  const u64 tile_end = tile_start + WGA_TILE < a.n_ops ? tile_start + WGA_TILE : a.n_ops;
  const u32 nt = (u32)(tile_end - tile_start);
  const u64 tile_cols = wave_get_u64(pre, 0);
  u32 gsh = a.no_table ? 8u : WGA_TBL_SHIFT;
  while ((tile_cols >> gsh) >= WGA_TBL_N) gsh++;
  const u64 re0 = wave_get_u64(pre, 12);
  const u32 kb0 = re0 < tile_end ? (u32)(re0 - tile_start) : 0xFFFFFFFFu;
  for (u32 k = tid; k < WGA_TBL_N + 2u; k += WGA_BLOCK) s_tbl[k] = 0u;
  if (tid < 2u) s_zero2[tid] = 0u;

  /* ---- phase A: 4 consecutive ops per thread, block scan into LDS (as in k_paf2maf_expand) ---- */
  {
    u32 opw[4];
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
    u32 cls[4];
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
    const u32 sv[4] = {sl, si, sd, cnt};
    u32 sx[4], stot[4];
    block_excl_scan4_u32(sv, sx, stot, s_w4, true);
    u32 x_col = sx[0], x_i = sx[1], x_d = sx[2], x_cnt = sx[3];
    s_pcol[tid] = (GT)x_col; /* prefix at the thread's first op: the planner rebuilds record boundaries from it */
    s_pcnt[tid] = x_cnt;
    for (int e = 0; e < 4; e++) {
      if (tid * 4u + (u32)e == kb0) { /* the tile's first record ends before this op */
        s_bnd0[0] = x_col;
        s_bnd0[1] = x_cnt;
      }
      const bool isi = cls[e] == CLS_I, isd = cls[e] == CLS_D;
      if (isi | isd) { /* ONE instance for both kinds of gap op */
        const u32 len = opw[e] >> 4;
        const u32 slot = isi ? (x_cnt & 0xFFFFu) : (x_cnt >> 16);
        GT* const g_col = isi ? s_tg_col : s_qg_col;
        GT* const g_cum = isi ? s_tg_cum : s_qg_cum;
        g_col[slot] = (GT)x_col;
        g_cum[slot] = (GT)(isi ? x_i : x_d);
        tbl_mark_event(s_tbl, x_col, len, gsh, isi ? 0u : 16u);
        x_i += isi ? len : 0u;
        x_d += isi ? 0u : len;
        x_cnt += isi ? 1u : 0x10000u;
      }
      x_col += l[e];
    }
    if (tid == WGA_BLOCK - 1) { /* sentinels: totals (two, so that index i+1 is always readable) */
      s_tot[0] = x_col;
      s_tot[1] = x_cnt;
      s_tg_col[x_cnt & 0xFFFFu] = (GT)x_col;
      s_tg_cum[x_cnt & 0xFFFFu] = (GT)x_i;
      s_tg_col[(x_cnt & 0xFFFFu) + 1u] = (GT)x_col;
      s_tg_cum[(x_cnt & 0xFFFFu) + 1u] = (GT)x_i;
      s_qg_col[x_cnt >> 16] = (GT)x_col;
      s_qg_cum[x_cnt >> 16] = (GT)x_d;
      s_qg_col[(x_cnt >> 16) + 1u] = (GT)x_col;
      s_qg_cum[(x_cnt >> 16) + 1u] = (GT)x_d;
    }
    __syncthreads();
    tbl_scan(s_tbl, s_w4);
  }

#if defined(WGA_P_ABLATE) && WGA_P_ABLATE == 1
  return;
#endif
  /* ---- phase B: wave 0 plans (pieces of up to a few record segments per round), every wave emits its pieces ---- */
  /* The planner's state lives in LDS between rounds (s_plan[2..6]: record, ops done, column / event prefix at the
   * segment start, segments done), so that nothing of it occupies registers while the pieces are emitted. */
  const u32 r0 = wave_get_u32(pre, 2);
  if (tid == 0u) {
    s_plan[2] = r0;
    s_plan[3] = 0u;
    s_plan[4] = 0u;
    s_plan[5] = 0u;
    s_plan[6] = 0u;
  }
  __syncthreads();
  for (;;) {
    if (wave == 0u) {
      u32 r = WGA_UNI32(s_plan[2]);
      u64 cur = tile_start + WGA_UNI32(s_plan[3]);
      u32 bnd_col = WGA_UNI32(s_plan[4]), bnd_ev = WGA_UNI32(s_plan[5]), nseg = WGA_UNI32(s_plan[6]);
      u64 re = r == r0 ? re0 : a.op_off[r + 1];
      u32 np_out = 0u;
      while (cur < tile_end && np_out + WGA_P_PIECE_MAX <= WGA_P_PIECES) {
        while (re <= cur) {
          r++;
          re = a.op_off[r + 1];
        }
        const bool is0 = r == r0;
        const u64 rs = is0 ? wave_get_u64(pre, 10) : a.op_off[r];
        const u64 seg_end = re < tile_end ? re : tile_end;
        const u32 kb = (u32)(seg_end - tile_start);
        u64 b_mx = 0, b_i = 0, b_d = 0;
        if (rs < tile_start) { /* only the tile's first record can continue from earlier tiles */
          b_mx = wave_get_u64(pre, 4);
          b_i = wave_get_u64(pre, 6);
          b_d = wave_get_u64(pre, 8);
        }
        const u64 cb = b_mx + b_i + b_d; /* record-relative column of the segment start */
        const u64 tb = b_mx + b_d;       /* target bases consumed before it              */
        const u64 qb = b_mx + b_i;       /* query bases consumed before it               */
        u32 dsc = pre;                   /* record geometry spread over lanes: layout of wga_tile_desc */
        if (!is0) {
          const u32* rp = (const u32*)(a.recs + r);
          dsc = 0u;
          if (lane >= 14u && lane < 32u) dsc = rp[lane - 14u];
          if (lane == 3u) dsc = rp[18];
        }
        const u64 t_src_len = wave_get_u64(dsc, 20), q_src_len = wave_get_u64(dsc, 24);
        u32 col_b, evb;
        if (kb == nt) {
          col_b = WGA_UNI32(s_tot[0]);
          evb = WGA_UNI32(s_tot[1]);
        } else if (nseg == 0u) { /* written in phase A */
          col_b = WGA_UNI32(s_bnd0[0]);
          evb = WGA_UNI32(s_bnd0[1]);
        } else { /* a further record ends inside the tile: the prefix at op kb from its thread's, at most three ops on */
          col_b = WGA_UNI32((u32)s_pcol[kb >> 2]);
          evb = WGA_UNI32(s_pcnt[kb >> 2]);
          for (u32 e = 0; e < (kb & 3u); e++) {
            const u32 op = a.ops[tile_start + (kb & ~3u) + e];
            const u32 cl = op_class(op & 15u);
            col_b += cl <= CLS_D ? (op >> 4) : 0u;
            evb += cl == CLS_I ? 1u : (cl == CLS_D ? 0x10000u : 0u);
          }
          col_b = WGA_UNI32(col_b);
          evb = WGA_UNI32(evb);
        }
        const u32 col_a = bnd_col;
        const u32 seg_cols = col_b - col_a;
        const u32 eva = bnd_ev;
        bnd_col = col_b;
        bnd_ev = evb;
        const int ia = (int)(eva & 0xFFFFu), ib = (int)(evb & 0xFFFFu);
        const int ja = (int)(eva >> 16), jb = (int)(evb >> 16);
        const u32 icum_a = WGA_UNI32((u32)s_tg_cum[ia]);
        const u32 dcum_a = WGA_UNI32((u32)s_qg_cum[ja]);

        /* String::insert_str panics when the insertion point is beyond the string (cigar.rs:507,513): an I (D) op
         * whose target (query) consumption so far exceeds the fetched slice.  Checked on the compact gap lists; the
         * exact op index is only worked out (serial rescan by the detecting lane) when that ever happens. */
        {
          bool pan = false;
#pragma clang loop vectorize(disable) unroll(disable)
          for (int i = ia + (int)lane; i < ib; i += 64)
            pan |= tb + (u64)((u32)s_tg_col[i] - col_a) - (u64)((u32)s_tg_cum[i] - icum_a) > t_src_len;
#pragma clang loop vectorize(disable) unroll(disable)
          for (int i = ja + (int)lane; i < jb; i += 64)
            pan |= qb + (u64)((u32)s_qg_col[i] - col_a) - (u64)((u32)s_qg_cum[i] - dcum_a) > q_src_len;
          if (pan) {
            u64* const panic_idx = (u64*)&a.diag[r].panic_op_idx;
            u64 tp = tb, qp = qb;
            for (u64 k = cur; k < seg_end; k++) {
              const u32 op = a.ops[k];
              const u32 c = op_class(op & 15u);
              const u64 len = op >> 4;
              if ((c == CLS_I && tp > t_src_len) || (c == CLS_D && qp > q_src_len)) {
                atomicMin(panic_idx, k - rs);
                break;
              }
              if (c == CLS_MX || c == CLS_D) tp += len;
              if (c == CLS_MX || c == CLS_I) qp += len;
            }
          }
        }

        /* Row jobs: 0/1 = this segment of the target / query row (rows end where a short slice ends);
         * 2/3 = once the record ends in this tile, what the slices hold beyond the CIGAR. */
        const bool rec_ends = seg_end == re;
        const u64 L = wave_get_u64(dsc, 30);
        const u32 neg = wave_get_u32(dsc, 3);
#pragma nounroll
        for (int job = 0; job < 4; job++) {
          const bool is_q = (job & 1) != 0, is_tail = job >= 2;
          if (is_tail && !rec_ends) break;
          if (is_tail && !(neg & (is_q ? 4u : 2u))) continue; /* wga_rec_desc::neg bits 1 / 2: no tail */
          const int q2 = is_q ? 2 : 0, q4 = is_q ? 4 : 0;
          const u64 gap_total = wave_get_u64(dsc, 26 + q2); /* I bases (target row) / D bases (query row) */
          const u64 src_len = is_q ? q_src_len : t_src_len;
          const u64 row_len = src_len + gap_total;
          u64 x0, nbytes;
          if (!is_tail) {
            const u64 x1 = cb + seg_cols < row_len ? cb + seg_cols : row_len;
            x0 = cb;
            nbytes = x1 > cb ? x1 - cb : 0;
          } else {
            x0 = L;
            nbytes = row_len > L ? row_len - L : 0;
          }
          if (nbytes == 0) continue;
          const u64 src_off = wave_get_u64(dsc, 18 + q4);
          const u64 fa_bytes = is_q ? a.q_fa_bytes : a.t_fa_bytes;
          const u8* const fa = is_q ? a.q_fa : a.t_fa;
          const bool rc = is_q && (neg & 1u) != 0u;
          const bool safe = src_off >= 16 && src_off + src_len + 16 <= fa_bytes; /* rowsrc_prepare */
          const u64 dst0 = (u64)(a.out) + wave_get_u64(dsc, 14 + q2) + x0;      /* address of the job's first byte */
          const u32 flags = (is_q ? 1u : 0u) | (rc ? 2u : 0u) | (safe ? 4u : 0u) | (is_tail ? 8u : 0u);
          if (is_tail) { /* rare (the PAF's coordinates disagree with its CIGAR): bytes, 2^30 at a time */
            const u64 m = nbytes < (1ull << 30) ? nbytes : (1ull << 30); /* longer tails: the owner loops (PD_N = 0 marks it) */
            piece_put(s_piece + np_out * 16u, lane, dst0, nbytes == m ? (u32)m : 0u, 0u, 0u, 0u, 0u, flags,
                      L - gap_total, nbytes, r);
            np_out++;
            continue;
          }
          const u64 sbase = is_q ? qb : tb;
          const u64 win = rc ? (u64)fa + src_off + src_len - 16 - sbase : (u64)fa + src_off + sbase; /* rowsrc_prepare */
          const u32 gab = is_q ? ((u32)ja | ((u32)jb << 16)) : ((u32)ia | ((u32)ib << 16));
          const u32 gcum = is_q ? dcum_a : icum_a;
          /* pieces: 1, 2 or 4, cut where the OUTPUT ADDRESS is a multiple of 128 (no line shared by two waves).
           * nbytes <= seg_cols < 2^31. */
          const u32 nb = (u32)nbytes;
          const u32 dst7 = (u32)dst0 & 127u;
          const u32 sh = nb <= WGA_K2P_SPLIT ? 0u : (nb <= 4u * WGA_K2P_SPLIT ? 1u : 2u);
          const u32 npc = 1u << sh;
          u32 lo = 0;
#pragma nounroll
          for (u32 p = 1; p <= npc; p++) {
            u32 hi = nb;
            if (p < npc) {
              hi = ((dst7 + (u32)(((u64)nb * p) >> sh) + 127u) & ~127u) - dst7;
              hi = hi < nb ? hi : nb;
            }
            if (hi <= lo) continue;
            piece_put(s_piece + np_out * 16u, lane, dst0 + lo, hi - lo, col_a + lo, col_a, gcum, gab, flags, sbase, win, r);
            np_out++;
            lo = hi;
          }
        }
        cur = seg_end;
        r++;
        if (cur < tile_end) re = a.op_off[r + 1];
        nseg++;
      }
      if (lane == 0u) {
        s_plan[0] = np_out;
        s_plan[1] = cur < tile_end ? 1u : 0u;
        s_plan[2] = r;
        s_plan[3] = (u32)(cur - tile_start);
        s_plan[4] = bnd_col;
        s_plan[5] = bnd_ev;
        s_plan[6] = nseg;
      }
    }
    __syncthreads();
    const u32 n_pieces = WGA_UNI32(s_plan[0]);
    const bool more = WGA_UNI32(s_plan[1]) != 0u;
    /* ---- emit: piece p belongs to wave p mod 4 ---- */
#pragma nounroll
    for (u32 p = wave; p < n_pieces; p += 4u) {
#if defined(WGA_P_ABLATE) && WGA_P_ABLATE == 2
      break;
#endif
      u32 pd = 0u;
      if (lane < 16u) pd = s_piece[p * 16u + lane];
      const u32 flags = wave_get_u32(pd, PD_FLAGS);
      const u32 rec = wave_get_u32(pd, PD_REC);
      u64* const bad_base = (u64*)&a.diag[rec].bad_base_pos;
      const bool is_q = (flags & 1u) != 0u;
      RowDescS<GT> rd;
      rd.c_org = wave_get_u32(pd, PD_CORG);
      rd.G_col = is_q ? s_qg_col : s_tg_col;
      rd.G_cum = rd.G_adj = is_q ? s_qg_cum : s_tg_cum;
      const u32 gab = wave_get_u32(pd, PD_GAB);
      rd.ga = (int)(gab & 0xFFFFu);
      rd.gb = (int)(gab >> 16);
      rd.gcum_a = wave_get_u32(pd, PD_GCUM);
      rd.sbase = wave_get_u64(pd, PD_SBASE);
      rd.lowmask = s_lowmask;
      rd.tbl = s_tbl;
      rd.tsh = is_q ? 16u : 0u;
      rd.gsh = gsh;
      rd.queue = s_queue + wave * WGA_P_PER;
      rd.res = s_res + wave * 64u;
      u8* const dst = (u8*)a.out + (wave_get_u64(pd, PD_DST) - (u64)a.out); /* pointer arithmetic on the kernel's own pointer */
      const u32 N = wave_get_u32(pd, PD_N), c_first = wave_get_u32(pd, PD_C0);
      if ((flags & 12u) == 4u) { /* safe, not a tail: the row emitter */
        const u8* const fa = is_q ? a.q_fa : a.t_fa;
        const u8* const win = fa + (i64)(wave_get_u64(pd, PD_WIN) - (u64)fa);
        if (flags & 2u)
          emit_piece_p<true, GT>(dst, N, c_first, rd, win, bad_base);
        else
          emit_piece_p<false, GT>(dst, N, c_first, rd, win, bad_base);
      } else { /* a row at a pool edge, or a slice tail: byte by byte */
        const wga_rec_desc rdsc = a.recs[rec];
        RowSrc src;
        src.fa = is_q ? a.q_fa : a.t_fa;
        src.fa_bytes = is_q ? a.q_fa_bytes : a.t_fa_bytes;
        src.src_off = is_q ? rdsc.q_src_off : rdsc.t_src_off;
        src.src_len = is_q ? rdsc.q_src_len : rdsc.t_src_len;
        src.rc = (flags & 2u) != 0u;
        src.ablate = 0;
        src.safe = false;
        src.win_base = nullptr;
        if (flags & 8u) { /* tail: no events; PD_WIN holds the byte count */
          rd.ga = rd.gb = 0;
          rd.gcum_a = 0u;
          rd.c_org = 0u;
          rd.tbl = s_zero2;
          rd.gsh = 31u;
          const u64 total = wave_get_u64(pd, PD_WIN), sb0 = rd.sbase;
          for (u64 done = 0; done < total; done += (1ull << 30)) {
            const u64 m = total - done < (1ull << 30) ? total - done : (1ull << 30);
            rd.sbase = sb0 + done;
            emit_bytes_p(dst + done, (u32)m, 0u, rd, src, bad_base);
          }
        } else {
          emit_bytes_p(dst, N, c_first, rd, src, bad_base);
        }
      }
    }
    if (!more) break;
    __syncthreads(); /* the piece list is rewritten by the next round */
  }
}

#define WGA_K2P_SHARED(GT)                                                                        \
  __shared__ u32 s_bnd0[2];                                                                       \
  __shared__ u32 s_tot[2];                                                                        \
  __shared__ u32 s_plan[8];                                                                       \
  __shared__ GT s_tg_col[WGA_TILE + 2];                                                           \
  __shared__ GT s_tg_cum[WGA_TILE + 2];                                                           \
  __shared__ GT s_qg_col[WGA_TILE + 2];                                                           \
  __shared__ GT s_qg_cum[WGA_TILE + 2];                                                           \
  __shared__ GT s_pcol[WGA_BLOCK];                                                                \
  __shared__ u32 s_pcnt[WGA_BLOCK];                                                               \
  __shared__ u32 s_zero2[2];                                                                      \
  __shared__ u32 s_w4[16];                                                                        \
  __shared__ u32 s_piece[WGA_P_PIECES * 16];                                                      \
  __shared__ u32x4_a16 s_lowmask[17];                                                             \
  __shared__ u32 s_tbl[WGA_TBL_N + 2];                                                            \
  __shared__ u16 s_queue[4 * WGA_P_PER];                                                          \
  __shared__ u32x4_a16 s_res[4 * 64];

/* narrow tiles: one block per tile of the batch; wide ones (and those that need the u64 walk) return at once */
__global__ __launch_bounds__(256, WGA_K2P_BLOCKS) void k_paf2maf_expand_p(ExpandArgsP a) {
  WGA_K2P_SHARED(u16)
  const u32 lane = threadIdx.x & 63u;
  const u64 g = xcd_tile_of_block();
  u32 pre = 0u;
  if (lane < 32u) pre = ((const u32*)(a.tdesc + g))[lane];
  const u64 tile_cols = wave_get_u64(pre, 0);
  if (tile_cols >= WGA_NARROW_COLS) return; /* block-uniform, before any barrier */
  build_lowmask(s_lowmask);
  expand_tile_p<u16>(a, g, pre, s_tg_col, s_tg_cum, s_qg_col, s_qg_cum, s_tbl, s_bnd0, s_tot, s_zero2, s_w4, s_pcol,
                     s_pcnt, s_plan, s_piece, s_lowmask, s_queue, s_res);
}

/* wide tiles (65 536 .. 2^31 columns), from the list k_list_wide_tiles wrote: u32 gap lists, fewer blocks per CU */
__global__ __launch_bounds__(256, 4) void k_paf2maf_expand_p_wide(ExpandArgsP a) {
  WGA_K2P_SHARED(u32)
  const u32 lane = threadIdx.x & 63u;
  const u32 n_wide = *a.wide_count;
  build_lowmask(s_lowmask);
  for (u32 idx = blockIdx.x; idx < n_wide; idx += gridDim.x) {
    const u64 g = a.wide_list[idx];
    u32 pre = 0u;
    if (lane < 32u) pre = ((const u32*)(a.tdesc + g))[lane];
    __syncthreads(); /* the previous tile's LDS state is dead */
    expand_tile_p<u32>(a, g, pre, s_tg_col, s_tg_cum, s_qg_col, s_qg_cum, s_tbl, s_bnd0, s_tot, s_zero2, s_w4, s_pcol,
                       s_pcnt, s_plan, s_piece, s_lowmask, s_queue, s_res);
  }
}

/* counts[0], list[0 .. nt): tiles of 65 536 .. 2^31 columns; counts[1], list[nt .. 2 nt): tiles beyond */
__global__ __launch_bounds__(256) void k_list_wide_tiles(const wga_tile_desc* descs, u64 nt, u32* counts,
                                                         u32* list) {
  const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
  if (g >= nt) return;
  const u64 cols = descs[g].tile_cols;
  if (cols < WGA_NARROW_COLS) return;
  if (cols <= WGA_FAST_COL_LIMIT)
    list[atomicAdd(&counts[0], 1u)] = (u32)g;
  else
    list[nt + atomicAdd(&counts[1], 1u)] = (u32)g;
}

#endif
