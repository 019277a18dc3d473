/*
 * wga_kernels_k2s.h — K2 `k_paf2maf_expand_s`: the paf2maf row kernel with LINE-COMPLETE stores.
 *
 * Why (scripts/micro/line_split_copy.hip, MI355X): when the 16-byte granules of one 128-byte output line
 * reach the L2 in different store instructions separated by a memory round trip — which is what a
 * "fast granules now, gap-touching granules from a queue later" emitter does to 60 % of its lines — the
 * L2 fills the partially written line from memory and writes it back more than once: a plain copy drops
 * from 5.2-5.6 TB/s to 2.8-3.0 TB/s, FETCH_SIZE doubles and WRITE_SIZE grows 1.4-1.5 x.  Assembling a
 * wave's output window in LDS first and storing it as whole lines costs 3-7 % of the copy rate.
 *
 * So: the tile analysis (phase A: gap lists + granule table) is the one of wga_kernels.h, but a row piece
 * is emitted window by window.  A window is WGA_STG_SPAN bytes of OUTPUT ADDRESS space, 128-byte aligned.
 * Per window the owning wave
 *   1. classifies the 64 x WGA_STG_U column granules that start inside the window, loads the plain ones
 *      (one byte-unaligned 16 B buffer load each, reverse-complement fused) and writes ALL granule slots of
 *      the stage — 16-byte aligned LDS writes, no predicate (slots of queued / inactive granules hold
 *      garbage until step 2 / are never flushed);
 *   2. drains the queue of gap-touching and row-edge granules of THIS window into their slots;
 *   3. flushes: every lane reads 16 B at the window's alignment (hardware-unaligned ds_read_b128) and
 *      stores them 16-byte aligned — 8 consecutive lanes write one whole 128-byte line in ONE instruction;
 *      only the first / last 16 bytes of a row piece are byte stores;
 *   4. carries the 1..15 bytes that the window's last granule holds beyond the window into the next one.
 * Rows are cut into pieces (1, 2 or 4, at 128-byte aligned output addresses) that go round-robin to the
 * block's four waves; there is no block-cooperative row path.
 *
 * Tiles narrower than 65 536 columns — all but pathological ones — keep their gap lists as u16: 22 KB of
 * LDS per block and <= 72 VGPRs give seven blocks per CU.  Wider tiles are listed by k_tile_base and run
 * through the u32 instance of the same code in a second, normally empty, launch.
 */
#ifndef WGA_KERNELS_K2S_H
#define WGA_KERNELS_K2S_H

#include "wga_kernels.h"

typedef unsigned short u16;

#ifndef WGA_STG_U
#define WGA_STG_U 2 /* column granules per lane and window */
#endif
#define WGA_STG_PER (64u * WGA_STG_U)        /* granule slots of a window                          */
#define WGA_STG_SPAN (WGA_STG_PER * 16u)     /* bytes of a window                                  */
#define WGA_STG_BYTES (WGA_STG_SPAN + 16u)   /* stage of a wave: 16-byte carry slot + granule slots */
#ifndef WGA_K2S_BLOCKS
#define WGA_K2S_BLOCKS 7
#endif
#ifndef WGA_K2S_SPLIT
#define WGA_K2S_SPLIT 6144u /* rows beyond this many bytes are cut in two pieces, beyond four times it in four */
#endif
#define WGA_NARROW_COLS 65536ull /* tiles below this many columns take the u16 instance */

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
  u16* queue; /* this wave's queue: WGA_STG_PER slot indices */
  u8* stage;  /* this wave's stage: WGA_STG_BYTES, 16-byte aligned */
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
 * windows under byte masks (the straight-line scheme of complex_chunk in wga_kernels.h); the sixteen bytes go
 * to the stage, bytes outside [g.a0, g.b0) are don't-care (the flush never writes them). */
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

/* The same granule in two steps, so that its loads are in flight together with the window's plain ones:
 * complex_addr_s works out the pieces and issues the two window loads, complex_finish_s merges.  Only what the
 * merge needs survives the wait — the four piece bounds as byte offsets inside the granule, packed in one word.
 * The rare continuations (an invalid base to report, a third event inside the sixteen columns) are left to the
 * caller, which redoes the granule with complex_granule_s. */
template <bool RC, typename GT>
__device__ __forceinline__ u32 complex_addr_s(const ChunkGeom& g, const RowDescS<GT>& rd, const RowBufs& rb,
                                              u32 r0[4], u32 r1[4]) {
  const int ga = rd.ga, gb = rd.gb;
  const u32 c = g.c, c_end = g.c_end, cz = g.cz;
  const int i = find_entry_s(rd, c);
  const bool has0 = i >= ga;
  const int ic = has0 ? i : ga;
  const u32 gs0 = rd.G_col[ic], cum0a = rd.G_cum[ic], cum0b = rd.G_cum[ic + 1];
  const u32 adj0b = rd.G_adj[ic + 1];
  const u32 gl0 = cum0b - cum0a;
  const bool in_gap0 = has0 && (c - gs0 < gl0);
  const u32 adj0 = has0 ? adj0b : rd.gcum_a;
  const int n1 = i + 1;
  const u32 gs1 = n1 < gb ? (u32)rd.G_col[n1] : 0xFFFFFFFFu;
  const u32 gl1 = (u32)rd.G_cum[n1 + 1] - (u32)rd.G_cum[n1];
  const u32 adj1 = rd.G_adj[n1 + 1];
  const u32 gs2 = n1 + 1 < gb ? (u32)rd.G_col[n1 + 1] : 0xFFFFFFFFu;
  const bool hasB = gs1 < c_end;
  const u32 g0e = gs0 + gl0;
  const u32 a1 = in_gap0 ? (g0e < c_end ? g0e : c_end) : c;
  const u32 b1 = hasB ? gs1 : c_end;
  const u32 g1e = gs1 + gl1;
  const u32 e1 = hasB ? (g1e < c_end ? g1e : c_end) : c_end;
  const u32 b2 = hasB ? (gs2 < c_end ? gs2 : c_end) : c_end;
  const int offz = (int)(cz - rd.c_org);
  const int off0 = offz - (int)(adj0 - rd.gcum_a), off1 = offz - (int)(adj1 - rd.gcum_a);
  buf_load16(rb.lbuf, b1 > a1 ? rowbuf_loff(rb, off0) : WGA_BUF_OOB, r0);
  buf_load16(rb.lbuf, b2 > e1 ? rowbuf_loff(rb, off1) : WGA_BUF_OOB, r1);
  return (a1 - cz) | ((b1 - cz) << 8) | ((e1 - cz) << 16) | ((b2 - cz) << 24);
}
/* returns true when the granule needs the serial redo */
template <bool RC>
__device__ __forceinline__ bool complex_finish_s(u32 meta, u32 b0, const u32x4_a16* lowmask, const u32 r0[4],
                                                 const u32 r1[4], u32 o[4]) {
  const u32 a1 = meta & 0xFFu, b1 = (meta >> 8) & 0xFFu, e1 = (meta >> 16) & 0xFFu, b2 = meta >> 24;
  const u32x4_a16 La1 = lowmask[a1], Lb1 = lowmask[b1], Le1 = lowmask[e1], Lb2 = lowmask[b2];
  u32 W0[4], W1[4], inv0[4], inv1[4];
  win_finish_t<RC>(r0, W0, inv0);
  win_finish_t<RC>(r1, W1, inv1);
  u32 bad = 0u;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const u32 m0 = Lb1[d] & ~La1[d], m1 = Lb2[d] & ~Le1[d];
    const u32 md = bfi32(Lb1[d], La1[d], Le1[d]);
    o[d] = bfi32(m0, W0[d], bfi32(m1, W1[d], md & 0x2D2D2D2Du));
    bad |= (inv0[d] & m0) | (inv1[d] & m1);
  }
  return (bool)((int)(RC && bad != 0u) | (int)(b2 < b0));
}

/* One row piece = N bytes at dst whose first byte is tile-relative column c0, emitted by ONE wave.  The row's
 * source windows must not need bounds checks (RowSrc::safe); win_base as rowsrc_prepare leaves it.
 * RC = the row is read reverse-complemented. */
template <bool RC, typename GT>
__device__ __forceinline__ void emit_piece_s(u8* dst, u32 N, u32 c0, const RowDescS<GT>& rd,
                                             const u8* win_base, u64* bad_base_pos) {
  const u32 lane = threadIdx.x & 63u;
  const RowGeom rg = row_geom(dst, N, c0);
  /* window geometry.  P0 = rg.base = address of byte 0 of granule 0 (<= dst), W0 = its 128-byte line.
   * Window k covers addresses [W0 + k SPAN, + SPAN) and owns the granules that start inside it: slot i of
   * window k is granule R0 + k PER + i (R0 <= 0: the first window's leading slots are empty).  Address of
   * stage byte s in window k: W0 + k SPAN + (s - e). */
  const u32 delta = (u32)((u64)rg.base & 127u);
  u8* const W0 = rg.base - delta;
  const int R0 = -(int)(delta >> 4);
  const u32 e = 16u - (delta & 15u);            /* stage offset of a window's first byte: 1..16 */
  const u32 offL = delta + rg.head, offH = offL + N; /* the piece's bytes, relative to W0 */
  /* windows by ADDRESS range: the piece's last granule may end in the window after the one it starts in */
  const u32 nwin = (offH + WGA_STG_SPAN - 1u) / WGA_STG_SPAN;
  u16* const queue = rd.queue;
  u8* const stage = rd.stage;
  const u32 lo_full = rg.head == 0u ? 0u : 1u;
  const u32 n_full = rg.nchunks - lo_full - (rg.last_b0 == 16u ? 0u : 1u); /* may wrap to "none" */
  const bool any_full = rg.nchunks >= lo_full + (rg.last_b0 == 16u ? 0u : 1u) + 1u;
  const int koff = (int)(rd.gcum_a - rd.c_org); /* window offset of a granule = cz + koff - adj */
  RowBufs rb;
  rb.sgn = RC ? 0xFFFFFFFFu : 0u;
  rb.kbias = RC ? 0x80000000u : 64u;
  rb.lbuf = buf_make(win_base - (i64)rb.kbias, 0xFFFFFFF0u);
  rb.sbuf = buf_make(W0, offH);
#pragma nounroll
  for (u32 k = 0; k < nwin; k++) {
    const int Rk = R0 + (int)(k * WGA_STG_PER);
    u32 qn = 0; /* wave-uniform queue length */
    /* ---- 1. classify; queue the granules that touch an event or a row edge; load the plain ones ---- */
    u32 loff[WGA_STG_U];
    bool dash[WGA_STG_U];
    u32 raw[WGA_STG_U][4];
#pragma unroll
    for (int u = 0; u < WGA_STG_U; u++) {
      const u32 rel = (u32)(Rk + (int)((u32)u * 64u + lane)); /* wraps for the empty leading slots */
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
      dash[u] = st == (WGA_TBL_COVER | WGA_TBL_FULL);
      /* bitwise, not &&: short-circuit evaluation would come back as exec-mask branches */
      const bool cand = (bool)((int)any_full & (int)(rel - lo_full < n_full) & (int)(((w0 ^ w1) & WGA_TBL_CNT) == 0u) &
                               (int)(st != WGA_TBL_COVER));
      const u32 off = cz + (u32)koff - adj; /* slice index of the granule relative to sbase, >= 0 */
      loff[u] = ((int)cand & (int)!dash[u]) ? rowbuf_loff(rb, (int)off) : WGA_BUF_OOB;
      const bool cx = (bool)((int)act & (int)!cand);
      const u64 m = __ballot(cx);
      if (cx) queue[qn + lane_rank(m, lane)] = (u16)((u32)u * 64u + lane);
      qn += (u32)__popcll(m);
    }
#pragma unroll
    for (int u = 0; u < WGA_STG_U; u++) buf_load16(rb.lbuf, loff[u], raw[u]);
    WGA_WAVE_SYNC();
    /* ---- 2a. the first 64 queued granules: pieces and window loads, in flight together with the plain ones ---- */
    const u32 take0 = qn < 64u ? qn : 64u;
    u32 c_meta = 0u, c_qi = 0u, c_r0[4] = {0u, 0u, 0u, 0u}, c_r1[4] = {0u, 0u, 0u, 0u};
    if (lane < take0) {
      c_qi = queue[qn - take0 + lane];
      c_meta = complex_addr_s<RC>(chunk_geom(rg, (u32)(Rk + (int)c_qi)), rd, rb, c_r0, c_r1);
    }
    /* ---- 1b. the plain granules arrive: every slot of the stage is written, no predicate ---- */
#pragma unroll
    for (int u = 0; u < WGA_STG_U; u++) {
      u32 o[4], inv[4];
      win_finish_t<RC>(raw[u], o, inv);
      if (RC && loff[u] != WGA_BUF_OOB && (inv[0] | inv[1] | inv[2] | inv[3]) != 0u) { /* InvalidBase (utils.rs:97): rare */
        const u32 x = loff[u] - rb.kbias;
        flag_bad_bases(inv, 0, 16, (i64)rd.sbase + (int)((x ^ rb.sgn) - rb.sgn), rd.lowmask, bad_base_pos);
      }
#pragma unroll
      for (int d = 0; d < 4; d++) o[d] = dash[u] ? 0x2D2D2D2Du : o[d];
      const u32x4_a16 ov = {o[0], o[1], o[2], o[3]};
      *(u32x4_a16*)(stage + 16u + ((u32)u * 64u + lane) * 16u) = ov;
    }
    WGA_WAVE_SYNC();
    /* ---- 2b. merge the queued granules into their slots ---- */
    if (lane < take0) {
      u32 o[4];
      const ChunkGeom g = chunk_geom(rg, (u32)(Rk + (int)c_qi));
      if (complex_finish_s<RC>(c_meta, g.b0, rd.lowmask, c_r0, c_r1, o))
        complex_granule_s<RC>(g, rd, rb, bad_base_pos, o);
      const u32x4_a16 ov = {o[0], o[1], o[2], o[3]};
      *(u32x4_a16*)(stage + 16u + c_qi * 16u) = ov;
    }
    qn -= take0;
    while (qn > 0u) { /* more than 64 of them in one window: indel-dense stretches */
      const u32 take = qn < 64u ? qn : 64u;
      qn -= take;
      if (lane < take) {
        const u32 qi = queue[qn + lane];
        u32 o[4];
        complex_granule_s<RC>(chunk_geom(rg, (u32)(Rk + (int)qi)), rd, rb, bad_base_pos, o);
        const u32x4_a16 ov = {o[0], o[1], o[2], o[3]};
        *(u32x4_a16*)(stage + 16u + qi * 16u) = ov;
      }
    }
    WGA_WAVE_SYNC();
    /* ---- 3. flush whole lines ---- */
    const u32 qk = k * WGA_STG_SPAN;
    bool part[WGA_STG_U];
    u32 fv[WGA_STG_U][4];
#pragma unroll
    for (int u = 0; u < WGA_STG_U; u++) {
      const u32 j = (u32)u * 64u + lane;
      const u32x4_a1 v = *(const u32x4_a1*)(stage + e + j * 16u);
      fv[u][0] = v[0];
      fv[u][1] = v[1];
      fv[u][2] = v[2];
      fv[u][3] = v[3];
      const u32 q = qk + j * 16u;
      const bool full = (bool)((int)(q >= offL) & (int)(q + 16u <= offH));
      part[u] = (bool)((int)!full & (int)(q + 16u > offL) & (int)(q < offH));
      buf_store16(rb.sbuf, full ? q : WGA_BUF_OOB, fv[u]);
    }
#pragma unroll
    for (int u = 0; u < WGA_STG_U; u++) {
      if (part[u]) { /* the first / last sixteen bytes of the piece: byte stores, never read-modify-write */
        const u32 q = qk + ((u32)u * 64u + lane) * 16u;
        const u32 lo = q < offL ? offL - q : 0u, hi = q + 16u > offH ? offH - q : 16u;
        u8* const p = W0 + q;
#pragma clang loop vectorize(disable) unroll(disable)
        for (u32 b = lo; b < hi; b++) {
          const u32 d = b >> 2;
          const u32 word = d == 0 ? fv[u][0] : d == 1 ? fv[u][1] : d == 2 ? fv[u][2] : fv[u][3];
          p[b] = (u8)(word >> (8u * (b & 3u)));
        }
      }
    }
    WGA_WAVE_SYNC();
    /* ---- 4. carry: what the last slot holds beyond this window opens the next one ---- */
    if (lane == 0u) *(u32x4_a16*)stage = *(const u32x4_a16*)(stage + WGA_STG_SPAN);
    WGA_WAVE_SYNC();
  }
}

/* The same N bytes one at a time, by one wave: rows at a pool edge (their windows would need bounds checks) and
 * what a slice holds beyond its CIGAR (ga == gb: no events).  Neither occurs in a consistent PAF whose
 * sequences sit inside the pool; correctness only. */
template <typename GT>
__device__ __forceinline__ void emit_bytes_s(u8* dst, u32 N, u32 c0, const RowDescS<GT>& rd, const RowSrc& src,
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

struct ExpandArgsS {
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

/* the tile itself; GT = u16 (tile_cols < 65 536) or u32 (anything up to WGA_FAST_COL_LIMIT) */
template <typename GT>
__device__ __forceinline__ void expand_tile_s(const ExpandArgsS& a, const u64 g, u32 pre, GT* s_tg_col,
                                              GT* s_tg_cum, GT* s_qg_col, GT* s_qg_cum, u32* s_tbl,
                                              u32 (*s_bnd)[2], u32* s_tot, u32* s_zero2, u32* s_w4,
                                              const u32x4_a16* s_lowmask, u16* s_queue, u8* s_stage) {
  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  const u64 tile_start = g * WGA_TILE;
  const u64 tile_end = tile_start + WGA_TILE < a.n_ops ? tile_start + WGA_TILE : a.n_ops;
  const u32 nt = (u32)(tile_end - tile_start);
  const u64 tile_cols = wave_get_u64(pre, 0);
  u32 gsh = a.no_table ? 8u : WGA_TBL_SHIFT;
  while ((tile_cols >> gsh) >= WGA_TBL_N) gsh++;
  const u32 r0 = wave_get_u32(pre, 2);
  const u64 re0 = wave_get_u64(pre, 12);
  const u32 kb0 = re0 < tile_end ? (u32)(re0 - tile_start) : 0xFFFFFFFFu;
  for (u32 k = tid; k < WGA_TBL_N + 2u; k += WGA_BLOCK) s_tbl[k] = 0u;
  if (tid < 2u) s_zero2[tid] = 0u;

  /* ---- phase A: 4 consecutive ops per thread, block scan into LDS (as in k_paf2maf_expand) ---- */
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
  u32 my_col = 0, my_cnt = 0;
  {
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
    my_col = x_col;
    my_cnt = x_cnt;
    for (int e = 0; e < 4; e++) {
      if (tid * 4u + (u32)e == kb0) { /* the tile's first record ends before this op */
        s_bnd[0][0] = x_col;
        s_bnd[0][1] = x_cnt;
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
  __syncthreads();

  /* ---- phase B: the record segments of this tile ---- */
  u32 r = r0;
  u64 cur = tile_start;
  u64 re = wave_get_u64(pre, 12);
  u32 bnd_col = 0u, bnd_ev = 0u, nseg = 0u; /* prefix at the start of the current segment */
  u32 njob = 0u;                           /* row pieces seen so far (same count in every wave) */
  while (cur < tile_end) {
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
    u64* const bad_base = (u64*)&a.diag[r].bad_base_pos;
    u64* const panic_idx = (u64*)&a.diag[r].panic_op_idx;

    u32 col_b, evb;
    if (kb == nt) {
      col_b = WGA_UNI32(s_tot[0]);
      evb = WGA_UNI32(s_tot[1]);
    } else if (nseg == 0u) { /* written in phase A */
      col_b = WGA_UNI32(s_bnd[0][0]);
      evb = WGA_UNI32(s_bnd[0][1]);
    } else { /* a further record ends inside the tile: the owner of op kb rebuilds its prefix */
      u32* const slot = s_bnd[1u + (nseg & 1u)];
      if (tid == (kb >> 2)) {
        u32 c = my_col, n = my_cnt;
        for (u32 e = 0; e < (kb & 3u); e++) {
          const u32 op = a.ops[tile_start + (kb & ~3u) + e];
          const u32 cl = op_class(op & 15u);
          c += cl <= CLS_D ? (op >> 4) : 0u;
          n += cl == CLS_I ? 1u : (cl == CLS_D ? 0x10000u : 0u);
        }
        slot[0] = c;
        slot[1] = n;
      }
      __syncthreads(); /* uniform: every thread walks the same segments */
      col_b = WGA_UNI32(slot[0]);
      evb = WGA_UNI32(slot[1]);
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

    /* String::insert_str panics when the insertion point is beyond the string (cigar.rs:507,513) */
    bool pan = false;
#pragma clang loop vectorize(disable) unroll(disable)
    for (int i = ia + (int)tid; i < ib; i += (int)WGA_BLOCK)
      pan |= tb + (u64)((u32)s_tg_col[i] - col_a) - (u64)((u32)s_tg_cum[i] - icum_a) > t_src_len;
#pragma clang loop vectorize(disable) unroll(disable)
    for (int i = ja + (int)tid; i < jb; i += (int)WGA_BLOCK)
      pan |= qb + (u64)((u32)s_qg_col[i] - col_a) - (u64)((u32)s_qg_cum[i] - dcum_a) > q_src_len;
    if (pan) {
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

    /* Row jobs: 0/1 = this segment of the target / query row (rows end where a short slice ends);
     * 2/3 = once the record ends in this tile, what the slices hold beyond the CIGAR. */
    const bool rec_ends = seg_end == re;
#pragma nounroll
    for (int job = 0; job < 4; job++) {
      const bool is_q = (job & 1) != 0, is_tail = job >= 2;
      const int q2 = is_q ? 2 : 0, q4 = is_q ? 4 : 0;
      const u64 gap_total = wave_get_u64(dsc, 26 + q2); /* I bases (target row) / D bases (query row) */
      const u64 L = wave_get_u64(dsc, 30);
      const u64 src_len = is_q ? q_src_len : t_src_len;
      const u64 row_len = src_len + gap_total;
      u64 x0, nbytes;
      if (!is_tail) {
        const u64 x1 = cb + seg_cols < row_len ? cb + seg_cols : row_len;
        x0 = cb;
        nbytes = x1 > cb ? x1 - cb : 0;
      } else {
        if (!rec_ends) continue;
        x0 = L;
        nbytes = row_len > L ? row_len - L : 0;
      }
      if (nbytes == 0) continue;
      if (is_tail) { /* rare (the PAF's coordinates disagree with its CIGAR): one wave, byte by byte */
        if ((njob++ & 3u) != wave) continue;
        RowSrc src;
        src.fa = is_q ? a.q_fa : a.t_fa;
        src.fa_bytes = is_q ? a.q_fa_bytes : a.t_fa_bytes;
        src.src_off = wave_get_u64(dsc, 18 + q4);
        src.src_len = src_len;
        src.rc = is_q && wave_get_u32(dsc, 3) != 0u;
        src.ablate = 0;
        RowDescS<GT> rd;
        rd.c_org = 0u;
        rd.G_col = rd.G_cum = rd.G_adj = s_tg_col;
        rd.ga = rd.gb = 0;
        rd.gcum_a = 0u;
        rd.lowmask = s_lowmask;
        rd.tbl = s_zero2;
        rd.tsh = 0u;
        rd.gsh = 31u;
        rd.queue = s_queue;
        rd.stage = s_stage;
        u8* const dst = a.out + wave_get_u64(dsc, 14 + q2) + x0;
        for (u64 done = 0; done < nbytes; done += (1ull << 30)) {
          const u64 m = nbytes - done < (1ull << 30) ? nbytes - done : (1ull << 30);
          rd.sbase = L - gap_total + done;
          emit_bytes_s(dst + done, (u32)m, 0u, rd, src, bad_base);
        }
        continue;
      }
      /* pieces of the row: 1, 2 or 4, cut where the OUTPUT ADDRESS is a multiple of 128 so that no line is shared
       * by two waves; every wave works the cuts out (they decide who owns what), only the owner reads the rest.
       * nbytes <= seg_cols < 2^31. */
      const u32 nb = (u32)nbytes;
      const u32 dst7 = ((u32)(u64)(a.out) + wave_get_u32(dsc, 14 + q2) + (u32)x0) & 127u; /* the segment's first address mod 128 */
      const u32 sh = nb <= WGA_K2S_SPLIT ? 0u : (nb <= 4u * WGA_K2S_SPLIT ? 1u : 2u);
      const u32 np = 1u << sh;
      u32 lo = 0;
#pragma nounroll
      for (u32 p = 1; p <= np; p++) {
        u32 hi = nb;
        if (p < np) {
          hi = ((dst7 + (u32)(((u64)nb * p) >> sh) + 127u) & ~127u) - dst7;
          hi = hi < nb ? hi : nb;
        }
        if (hi <= lo) continue;
        const u32 plo = lo;
        lo = hi;
        if ((njob++ & 3u) != wave) continue;
        RowSrc src;
        src.fa = is_q ? a.q_fa : a.t_fa;
        src.fa_bytes = is_q ? a.q_fa_bytes : a.t_fa_bytes;
        src.src_off = wave_get_u64(dsc, 18 + q4);
        src.src_len = src_len;
        src.rc = is_q && wave_get_u32(dsc, 3) != 0u;
        src.ablate = 0;
        RowDescS<GT> rd;
        rd.c_org = col_a;
        rd.G_col = is_q ? s_qg_col : s_tg_col;
        rd.G_cum = rd.G_adj = is_q ? s_qg_cum : s_tg_cum;
        rd.ga = is_q ? ja : ia;
        rd.gb = is_q ? jb : ib;
        rd.gcum_a = is_q ? dcum_a : icum_a;
        rd.sbase = is_q ? qb : tb;
        rd.lowmask = s_lowmask;
        rd.tbl = s_tbl;
        rd.tsh = is_q ? 16u : 0u;
        rd.gsh = gsh;
        rd.queue = s_queue + wave * WGA_STG_PER;
        rd.stage = s_stage + wave * WGA_STG_BYTES;
        rowsrc_prepare(src, rd.sbase);
        u8* const dst = a.out + wave_get_u64(dsc, 14 + q2) + x0 + plo;
        if (!src.safe)
          emit_bytes_s(dst, hi - plo, col_a + plo, rd, src, bad_base);
        else if (src.rc)
          emit_piece_s<true, GT>(dst, hi - plo, col_a + plo, rd, src.win_base, bad_base);
        else
          emit_piece_s<false, GT>(dst, hi - plo, col_a + plo, rd, src.win_base, bad_base);
      }
    }
    cur = seg_end;
    r++;
    if (cur < tile_end) re = a.op_off[r + 1];
    nseg++;
  }
}

#define WGA_K2S_SHARED(GT)                                                                        \
  __shared__ u32 s_bnd[3][2];                                                                     \
  __shared__ u32 s_tot[2];                                                                        \
  __shared__ GT s_tg_col[WGA_TILE + 2];                                                           \
  __shared__ GT s_tg_cum[WGA_TILE + 2];                                                           \
  __shared__ GT s_qg_col[WGA_TILE + 2];                                                           \
  __shared__ GT s_qg_cum[WGA_TILE + 2];                                                           \
  __shared__ u32 s_zero2[2];                                                                      \
  __shared__ u32 s_w4[16];                                                                        \
  __shared__ u32x4_a16 s_lowmask[17];                                                             \
  __shared__ u32 s_tbl[WGA_TBL_N + 2];                                                            \
  __shared__ u16 s_queue[4 * WGA_STG_PER];                                                        \
  __shared__ __attribute__((aligned(16))) u8 s_stage[4 * WGA_STG_BYTES];

/* narrow tiles: one block per tile of the batch; wide ones (and those that need the u64 walk) return at once */
__global__ __launch_bounds__(256, WGA_K2S_BLOCKS) void k_paf2maf_expand_s(ExpandArgsS a) {
  WGA_K2S_SHARED(u16)
  const u32 lane = threadIdx.x & 63u;
  const u64 g = blockIdx.x;
  u32 pre = 0u;
  if (lane < 32u) pre = ((const u32*)(a.tdesc + g))[lane];
  const u64 tile_cols = wave_get_u64(pre, 0);
  if (tile_cols >= WGA_NARROW_COLS) return; /* block-uniform, before any barrier */
  build_lowmask(s_lowmask);
  expand_tile_s<u16>(a, g, pre, s_tg_col, s_tg_cum, s_qg_col, s_qg_cum, s_tbl, s_bnd, s_tot, s_zero2, s_w4,
                     s_lowmask, s_queue, s_stage);
}

/* wide tiles (65 536 .. 2^31 columns), from the list k_tile_base wrote: u32 gap lists, fewer blocks per CU */
__global__ __launch_bounds__(256, 4) void k_paf2maf_expand_s_wide(ExpandArgsS a) {
  WGA_K2S_SHARED(u32)
  const u32 lane = threadIdx.x & 63u;
  const u32 n_wide = *a.wide_count;
  build_lowmask(s_lowmask);
  for (u32 idx = blockIdx.x; idx < n_wide; idx += gridDim.x) {
    const u64 g = a.wide_list[idx];
    u32 pre = 0u;
    if (lane < 32u) pre = ((const u32*)(a.tdesc + g))[lane];
    __syncthreads(); /* the previous tile's LDS state is dead */
    expand_tile_s<u32>(a, g, pre, s_tg_col, s_tg_cum, s_qg_col, s_qg_cum, s_tbl, s_bnd, s_tot, s_zero2, s_w4,
                       s_lowmask, s_queue, s_stage);
  }
}

/* k_tile_base's companion: list the tiles the u16 instance leaves out */
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
