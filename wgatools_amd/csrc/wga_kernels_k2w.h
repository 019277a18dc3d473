/*
 * wga_kernels_k2w.h — K2 `k_paf2maf_expand_w`: the paf2maf row kernel (parse_cigar_to_insert + cigar_unit_insert_seq,
 * cigar.rs:492-551, with reverse_complement, utils.rs:83-101, fused) built around OUTPUT WINDOWS.
 *
 * What the first two kernels taught (profiles/r02_k2_experiments.md): v1 classifies every 16-column granule (two table
 * words + an adjustment, ~50 lane instructions per granule), queues the ones that touch a gap and emits them later —
 * it is VALU-issue bound, its lanes are half empty in the queue drains, and 60 % of its 128-byte output lines reach
 * the L2 in two pieces (1.4 x the write traffic, partial-line fills on top).  This kernel turns the work around:
 *
 *   * the unit is a WINDOW: WGA_W_BYTES of OUTPUT ADDRESS space of one row piece, 128-byte aligned, owned by one wave
 *     and assembled in LDS (the wave's stage).  Granules are 16-byte aligned ADDRESSES, so stage writes, stage reads
 *     and global stores are all aligned, and a window leaves as whole lines in one pass: no line is written twice.
 *   * plain fill: every granule of the window is loaded as if no gap touched it — source offset = column minus the
 *     gap bases of the events that START before the granule, read from a per-window table the wave builds with one
 *     LDS atomic per event and one wave scan (no per-granule classification, no queue: 1 LDS read, 1 subtract,
 *     1 window load, 1 stage write per granule).  Bytes behind a gap that starts inside a granule, and the dashes
 *     themselves, are wrong at this point.
 *   * fix-ups are EVENT-centric and dense: one lane per gap op of the window.  Pass 1: the first event that starts in
 *     a granule rebuilds that granule (old bytes | dashes | window behind the gap | a second event), pass 2: the
 *     dashes a gap carries into the granule where it ends, pass 3: whole-dash granules of long gaps, written by the
 *     wave together.  The passes are separate instruction sequences of one wave, each lane of a pass owns a distinct
 *     granule: no write conflicts.
 *   * the tile's windows (both rows of every record segment, and the slice tails) form one flat sequence that the four
 *     waves take round robin, so short records do not leave waves idle.
 *   * record boundaries inside the tile: every wave loads the next 64 `op_off` values with one coalesced load and
 *     rebuilds the (column, gap-op count) prefix at each boundary from a per-thread prefix array in LDS — no block
 *     barrier per record as in v1.
 *   * the two gap lists share one array pair: target-row events (I ops), their sentinels, then the query-row events
 *     (D ops) — I + D <= 1024 ops per tile: 8 KB of LDS instead of 16.
 *
 * Tiles beyond 2^31 columns and the "expand_force_slow" knob take v1's op-serial walk (k_paf2maf_expand_list); rows at
 * a pool edge (their window loads would need bounds checks) are written byte by byte (emit_span_bytes).
 */
#ifndef WGA_KERNELS_K2W_H
#define WGA_KERNELS_K2W_H

#include "wga_kernels.h"

#ifndef WGA_W_U
#define WGA_W_U 4 /* granules per lane and window */
#endif
#define WGA_W_G (64u * WGA_W_U)     /* granules of a window  */
#define WGA_W_BYTES (WGA_W_G * 16u) /* bytes of a window     */
#ifndef WGA_K2W_BLOCKS
#define WGA_K2W_BLOCKS 4 /* 128 VGPRs, no scratch; five blocks (96 VGPRs) spill and lose 25 % */
#endif
#ifndef WGA_AUTO_SHORT_OPS
#define WGA_AUTO_SHORT_OPS 100ull /* batches below this many ops per record take the window kernel when "expand_variant" is -1 (same buffers, round 4: 30-op records 6.1 against 7.3 ms for the streaming kernel, 60-op 7.0 / 7.4, 100-op 8.0 / 8.0, 200-op 8.1 / 6.9) */
#endif

/* The kernel's arguments for the code behind phase A: only the planner and the rare paths read them, so they are
 * fetched from the kernarg segment where they are used (scalar loads) instead of occupying ~40 SGPRs for the whole
 * kernel — the window loop would spill its own scalars to make room for them. */
typedef const WGA_KARG_SPACE ExpandArgs* KArgP;
#define WGA_KARG_PTR(a) WGA_KARG_SEGMENT(KArgP, a)


/* one row's gap events of the tile, in column order: col[e] = start column (tile relative), cum[e] = gap bases of the
 * entries before e; two sentinel entries behind the last one (column = the tile's width) */
struct EvList {
  const u32* col;
  const u32* cum;
  __device__ __forceinline__ u32 c(int e) const { return col[e]; }
  __device__ __forceinline__ u32 m(int e) const { return cum[e]; }
};

/* first e in [lo, hi) whose column is >= x (as signed values: x may be negative), hi if none.  Wave-uniform: 64-ary
 * search, one LDS probe + ballot per level; probes are clamped instead of predicated (no exec-mask branches). */
__device__ __forceinline__ int ev_first_ge(const EvList& L, int lo, int hi, int x, u32 lane) {
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) >> 6;
    const int p = lo + (int)(lane + 1u) * step - 1; /* last entry of this lane's run */
    const int pc = p < hi ? p : hi - 1;
    const int v = (int)L.c(pc);
    const int k = (int)__popcll(__ballot((int)(p < hi) & (int)(v < x)));
    lo = (int)WGA_UNI32((u32)(lo + k * step));
    const int nh = lo + step;
    hi = nh < hi ? nh : hi;
  }
  if (hi <= lo) return lo;
  const int p = lo + (int)lane;
  const int pc = p < hi ? p : hi - 1;
  const int v = (int)L.c(pc);
  return (int)WGA_UNI32((u32)(lo + (int)__popcll(__ballot((int)(p < hi) & (int)(v < x)))));
}

/* everything a wave needs to emit the windows of one row piece (a record segment's part of a row, or a slice tail) */
struct SpanW {
  u8* dst;      /* address of the piece's first byte                                          */
  u64 N;        /* bytes (a segment's piece: < 2^31; only a slice tail can be longer)          */
  int c0;       /* tile-relative column of the first byte (a tail: 0)                          */
  u32 c_org;    /* column at which the slice index is `sbase` (the segment's first column)     */
  EvList ev;    /* the row's events                                                            */
  int ea, eb;   /* ... of this segment: [ea, eb) (a tail: empty)                               */
  u32 cum_a;    /* ev.m(ea)                                                                    */
  BufRsrc lbuf; /* source windows relative to win_base, range checked (see plan_round)         */
  /* what only the rare paths need (InvalidBase rescans, rows at a pool edge): filled by span_rare */
  u64 sbase;    /* slice index of column c_org                                                 */
  u64 src_len;  /* slice length                                                                */
  const u8* slice; /* the slice's first byte in the pool (forward strand)                      */
  u64* bad_base_pos;
};


/* sixteen bytes: `a` where the mask is set, `b` elsewhere */
__device__ __forceinline__ void sel16(u32 o[4], const u32x4_a16& m, const u32 a[4], const u32 b[4]) {
#pragma unroll
  for (int d = 0; d < 4; d++) o[d] = bfi32(m[d], a[d], b[d]);
}

/* buffer offset of the source window at slice offset `off`: forward rows are biased by 16 bytes, reversed rows walk
 * down from 2^31 (the buffer of the piece is set up accordingly) */
template <bool RC>
__device__ __forceinline__ u32 w_loff(int off) {
  return RC ? 0x80000000u - (u32)off : (u32)off + 16u;
}

/* sixteen source bytes as the row wants them: as they are, or reversed + complemented with the invalid-base flags
 * (non-zero byte) of comp4 */
template <bool RC>
__device__ __forceinline__ void w_finish(const u32 r[4], u32 W[4], u32 inv[4]) {
  if (RC) {
    W[0] = comp4(bswap32(r[3]), &inv[0]);
    W[1] = comp4(bswap32(r[2]), &inv[1]);
    W[2] = comp4(bswap32(r[1]), &inv[2]);
    W[3] = comp4(bswap32(r[0]), &inv[3]);
  } else {
    W[0] = r[0], W[1] = r[1], W[2] = r[2], W[3] = r[3];
    inv[0] = inv[1] = inv[2] = inv[3] = 0u;
  }
}

/* One granule from an arbitrary state, any number of events (the rare continuation of the straight-line pass): bytes
 * [c, z + 16) of the granule at column z, event i the last one that starts at or before c.  Returns non-zero when a
 * byte it used is not a base (reverse-complemented rows). */
template <bool RC>
__device__ __forceinline__ u32 fix_walk(u32 o[4], u32 c, u32 z, int i, const SpanW& sp, const u32x4_a16* lowmask) {
  const u32 c_end = z + 16u;
  bool in_gap = false;
  u32 gap_end = 0u, bad = 0u;
  while (c < c_end) {
    if (in_gap) {
      const u32 pe = gap_end < c_end ? gap_end : c_end;
      merge_dash(o, (int)(c - z), (int)(pe - z), lowmask);
      c = pe;
      in_gap = false;
    } else {
      const u32 next_gs = (i + 1 < sp.eb) ? sp.ev.c(i + 1) : 0xFFFFFFFFu;
      const u32 pe = next_gs < c_end ? next_gs : c_end;
      if (pe > c) {
        const int pa = (int)(c - z), pb = (int)(pe - z);
        const int off = (int)(z - sp.c_org) - (int)(sp.ev.m(i + 1) - sp.cum_a);
        u32 raw[4], W[4], inv[4];
        buf_load16(sp.lbuf, w_loff<RC>(off), raw);
        w_finish<RC>(raw, W, inv);
        const u32x4_a16 hi = lowmask[pb], lo = lowmask[pa];
#pragma unroll
        for (int d = 0; d < 4; d++) bad |= inv[d] & hi[d] & ~lo[d];
        merge16(o, W, pa, pb, lowmask);
        c = pe;
      }
      if (c < c_end) { /* c == start of event i + 1 */
        i++;
        const u32 gl = sp.ev.m(i + 1) - sp.ev.m(i);
        if (gl) {
          in_gap = true;
          gap_end = sp.ev.c(i) + gl;
        }
      }
    }
  }
  return bad;
}

/* InvalidBase (utils.rs:97), the exact and rare part: some byte a lane used for granule `g` of the window was not a base
 * (or was a zero the range-checked load returned).  The granule's columns are walked one by one against the event list
 * and the slice itself is read: every column of the piece that maps to a slice byte other than ACGTNacgtn reports its
 * position; atomicMin keeps the first in reversed order.  Runs once per invalid base of the input. */
__device__ __forceinline__ void rescan_granule(const SpanW& sp, int cw0, u32 g, u32 vlo, u32 vhi) {
  for (u32 j = 0; j < 16u; j++) {
    const u32 t = (g << 4) + j;
    if (t < vlo || t >= vhi) continue;
    const u32 c = (u32)(cw0 + (int)t);
    int lo = sp.ea, hi = sp.eb; /* lo = events of the piece that start at or before c */
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (sp.ev.c(mid) <= c)
        lo = mid + 1;
      else
        hi = mid;
    }
    int i = lo - 1;
    while (i >= sp.ea && sp.ev.m(i + 1) == sp.ev.m(i)) i--;
    if (i >= sp.ea && c - sp.ev.c(i) < sp.ev.m(i + 1) - sp.ev.m(i)) continue; /* a dash */
    const u64 pos = sp.sbase + (u64)(c - sp.c_org) - (u64)(sp.ev.m(lo) - sp.cum_a);
    if (pos >= sp.src_len) continue;
    const u32 ch = sp.slice[sp.src_len - 1u - pos] & 0xDFu; /* upper case */
    if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'N') atomicMin(sp.bad_base_pos, pos);
  }
}

/* pass 1 of the fix-ups in two halves, so that its two window loads are in flight together with the plain fill's:
 * fix1_issue works out the pieces of the granule an event starts in and issues the loads, fix1_finish merges.
 * Layout of the granule at column z:  [0, a) what the plain fill wrote | [a, b) dashes of event e | [b, c) window behind e
 * | [c, d) dashes of event e + 1 | [d, 16) window behind e + 1;  a third event inside the sixteen columns is left to fix_walk. */
struct Fix1 {
  u32 abcd;    /* a | b << 8 | c << 16 | d << 24 (byte offsets inside the granule) */
  int g0;      /* the granule (window relative) */
  u32 r1[4], r2[4];
  bool own, third;
};
/* What a lane knows about the event it holds: read from the lists in ONE batch of LDS loads (no dependent round trips).
 * Entries behind the piece's last event are readable (the next segment's events or the sentinels) and are masked by `eb`. */
struct EvView {
  u32 cprev, gs, gsB, gsC; /* start columns of events e - 1, e, e + 1, e + 2 */
  u32 m0, m1, m2;          /* gap bases in front of events e, e + 1, e + 2   */
};
__device__ __forceinline__ void ev_view(EvView& v, const EvList& L, int e) {
  const int em = e > 0 ? e - 1 : 0;
  v.cprev = L.c(em);
  v.gs = L.c(e);
  v.gsB = L.c(e + 1);
  v.gsC = L.c(e + 2);
  v.m0 = L.m(e);
  v.m1 = L.m(e + 1);
  v.m2 = L.m(e + 2);
  WGA_PIN7(v.cprev, v.gs, v.gsB, v.gsC, v.m0, v.m1, v.m2); /* seven loads, one wait */
}
template <bool RC>
__device__ __forceinline__ void fix1_issue(Fix1& f, const SpanW& sp, const EvView& v, int cw0, int e_lo, int e, bool act) {
  const u32 gs = v.gs, cum1 = v.m1;
  const u32 ge = gs + (cum1 - v.m0);
  f.g0 = (int)((u32)((int)gs - cw0) >> 4);
  /* the first event that starts in a granule owns it */
  f.own = (bool)((int)act & ((int)(e == e_lo) | (int)((int)((u32)((int)v.cprev - cw0) >> 4) != f.g0)));
  const u32 z = (u32)(cw0 + (f.g0 << 4)), zend = z + 16u;
  const u32 gsB = e + 1 < sp.eb ? v.gsB : 0xFFFFFFFFu;
  const u32 cumB1 = v.m2;
  const u32 gsC = e + 2 < sp.eb ? v.gsC : 0xFFFFFFFFu;
  const bool inB = gsB < zend;
  const u32 b = ge < zend ? ge : zend;
  const u32 c = inB ? gsB : zend;
  const u32 geB = gsB + (cumB1 - cum1);
  const u32 d = inB ? (geB < zend ? geB : zend) : zend;
  f.third = (bool)((int)f.own & (int)inB & (int)(d < zend) & (int)(gsC < zend));
  const int offz = (int)(z - sp.c_org);
  const int off1 = offz - (int)(cum1 - sp.cum_a), off2 = offz - (int)(cumB1 - sp.cum_a);
  buf_load16(sp.lbuf, ((int)f.own & (int)(c > b)) ? w_loff<RC>(off1) : WGA_BUF_OOB, f.r1);
  buf_load16(sp.lbuf, ((int)f.own & (int)inB & (int)(d < zend)) ? w_loff<RC>(off2) : WGA_BUF_OOB, f.r2);
  f.abcd = (gs - z) | ((b - z) << 8) | ((c - z) << 16) | ((d - z) << 24);
}
/* returns non-zero when a byte it used is not a base */
template <bool RC>
__device__ __forceinline__ u32 fix1_finish(const Fix1& f, const SpanW& sp, int cw0, int e, u32x4_a16* stage,
                                           const u32x4_a16* lowmask) {
  u32 bad = 0u;
  if (f.own) {
    const u32 dashw[4] = {0x2D2D2D2Du, 0x2D2D2D2Du, 0x2D2D2D2Du, 0x2D2D2D2Du};
    const u32x4_a16 oldv = stage[f.g0];
    const u32 old[4] = {oldv[0], oldv[1], oldv[2], oldv[3]};
    const u32x4_a16 La = lowmask[f.abcd & 0xFFu], Lb = lowmask[(f.abcd >> 8) & 0xFFu], Lc = lowmask[(f.abcd >> 16) & 0xFFu],
                    Ld = lowmask[f.abcd >> 24];
    u32 W1[4], W2[4], inv1[4], inv2[4];
    w_finish<RC>(f.r1, W1, inv1);
    w_finish<RC>(f.r2, W2, inv2);
    u32 o[4], t1[4], t2[4];
    sel16(t2, Ld, dashw, W2); /* [c, d) dashes | [d, 16) window 2 (below c: overwritten next) */
    sel16(t1, Lc, W1, t2);    /* [b, c) window 1                                               */
    sel16(t2, Lb, dashw, t1); /* [a, b) dashes                                                 */
    sel16(o, La, old, t2);    /* what the plain fill wrote in front of the event               */
    if (RC) {
#pragma unroll
      for (int k = 0; k < 4; k++) bad |= (inv1[k] & Lc[k] & ~Lb[k]) | (inv2[k] & ~Ld[k]);
    }
    if (f.third) { /* a third event inside sixteen columns: rare, generic walk from where it starts */
      const u32 z = (u32)(cw0 + (f.g0 << 4));
      bad |= fix_walk<RC>(o, sp.ev.c(e + 2), z, e + 1, sp, lowmask);
    }
    const u32x4_a16 ov = {o[0], o[1], o[2], o[3]};
    stage[f.g0] = ov;
  }
  return bad;
}

/* passes 2 and 3 for the event a lane holds: the dashes a gap carries into the granule where it ends (read-modify-write of
 * that granule; no two gaps end in the same one), and the whole-dash granules of long gaps, written by the wave together */
__device__ __forceinline__ void fix23(const EvView& v, int cw0, bool act, bool is_cover, u32x4_a16* stage,
                                      const u32x4_a16* lowmask, u32 lane) {
  const u32 dashw[4] = {0x2D2D2D2Du, 0x2D2D2D2Du, 0x2D2D2D2Du, 0x2D2D2D2Du};
  const int cw_end = cw0 + (int)WGA_W_BYTES;
  const u32 gs = v.gs, gl = v.m1 - v.m0, ge = gs + gl;
  const int g0 = is_cover ? -1 : (int)((u32)((int)gs - cw0) >> 4);
  const u32 ge_clip = (int)ge < cw_end ? ge : (u32)cw_end;
  const int G1 = (int)((u32)((int)ge_clip - 1 - cw0) >> 4); /* granule of the gap's last dash inside the window */
  const bool tail = (bool)((int)act & (int)(gl != 0u) & (int)((int)ge > cw0) & (int)(G1 > g0));
  if (tail) {
    const u32 n = ge_clip - (u32)(cw0 + (G1 << 4)); /* 1 .. 16 */
    const u32x4_a16 oldv = stage[G1];
    const u32 old[4] = {oldv[0], oldv[1], oldv[2], oldv[3]};
    u32 o[4];
    sel16(o, lowmask[n], dashw, old);
    const u32x4_a16 ov = {o[0], o[1], o[2], o[3]};
    stage[G1] = ov;
  }
  u64 m = __ballot((int)tail & (int)(G1 - g0 >= 2));
  while (m) {
    const u32 src = (u32)__builtin_ctzll(m);
    const int lo = (int)wave_get_u32_dyn((u32)g0, src) + 1, hi = (int)wave_get_u32_dyn((u32)G1, src);
    const u32x4_a16 dv = {dashw[0], dashw[1], dashw[2], dashw[3]};
    for (int g = lo + (int)lane; g < hi; g += 64) stage[g] = dv;
    m &= m - 1;
  }
}

/* ---- the plan of a tile: one record per row piece, written to LDS by the planner wave ------------------------------ */
#define WGA_W_SPAN_WORDS 20u
#define WGA_W_MAXSPAN 64u /* pieces of one planning round: 16 record segments x (two rows + two slice tails) */
#define WGA_W_SEGS 16u
/* words of a piece record */
#define WSP_DST 0    /* 0, 1: address of the piece's first byte             */
#define WSP_NLO 2    /* bytes, low word (high word: WSP_NHI)                 */
#define WSP_C0 3     /* tile-relative column of the first byte (tail: 0)     */
#define WSP_CORG 4   /* column of slice index sbase                          */
#define WSP_EA 5     /* events [ea, eb) of the row's list                    */
#define WSP_EB 6
#define WSP_CUMA 7   /* gap bases of the row's events before ea              */
#define WSP_EVBASE 8 /* 0: target-row list, q_base: query-row list           */
#define WSP_LBASE 9  /* 9, 10: base address of the source-window buffer      */
#define WSP_LNUM 11  /* its range                                            */
#define WSP_FLAGS 12 /* WSF_* */
#define WSP_REC 13   /* record index                                         */
#define WSP_KA 14    /* the segment's ops [ka, kb), tile relative            */
#define WSP_KB 15
#define WSP_SBASE 16 /* 16, 17: slice index of column c_org                  */
#define WSP_W0 18    /* index of the piece's first window in the round       */
#define WSP_NHI 19
#define WSF_RC 1u     /* read reversed + complemented                                                            */
#define WSF_BYTES 2u  /* byte by byte: a slice at a pool edge, or a record whose slices are shorter than its CIGAR */
#define WSF_Q 4u      /* query row                                                                                */
#define WSF_TAIL 8u   /* what a slice holds beyond its CIGAR (no events)                                          */
#define WSF_PANIC 16u /* check the row's gap ops for String::insert_str beyond the end (cigar.rs:507,513)         */

#ifndef WGA_K2W_PANIC_FLAG
#define WGA_K2W_PANIC_FLAG 8u /* wga_rec_desc::neg bit 3: a slice is shorter than the CIGAR consumes (insert_str may panic) */
#endif

/* the rare-path fields of a piece, from its record descriptor */
__device__ __forceinline__ void span_rare(SpanW& sp, KArgP a, const u32* rec) {
  WGA_KARG_FRESH(a);
  const u32 r = rec[WSP_REC], flags = rec[WSP_FLAGS];
  const wga_rec_desc* rp = a->recs + r;
  const bool is_q = (flags & WSF_Q) != 0u;
  sp.sbase = (u64)rec[WSP_SBASE] | ((u64)rec[WSP_SBASE + 1] << 32);
  sp.src_len = is_q ? rp->q_src_len : rp->t_src_len;
  sp.slice = (is_q ? a->q_fa : a->t_fa) + (is_q ? rp->q_src_off : rp->t_src_off);
  sp.bad_base_pos = (u64*)&a->diag[r].bad_base_pos;
}

/* One window of a row piece: window w covers the addresses [B, B + WGA_W_BYTES), B = (dst & ~127) + w * WGA_W_BYTES.
 * `rec` = the piece's record in LDS. */
template <bool RC>
__device__ __forceinline__ void emit_window(KArgP a, const u32* rec, u32 w, const u32* s_col, const u32* s_cum,
                                            u32x4_a16* stage, const u32x4_a16* lowmask, u32 lane
) {
  u32* const wt = (u32*)stage; /* the adjustment table lives in the stage until the fill starts */
  /* the piece's record: one LDS load (lane k = word k), fields by v_readlane */
  const u32 rv = rec[lane < WGA_W_SPAN_WORDS ? lane : 0u];
  SpanW sp;
  {
    const u32 evbase = wave_get_u32(rv, WSP_EVBASE);
    sp.ev.col = s_col + evbase;
    sp.ev.cum = s_cum + evbase;
  }
  sp.dst = (u8*)wave_get_u64(rv, WSP_DST);
  sp.N = (u64)wave_get_u32(rv, WSP_NLO) | ((u64)wave_get_u32(rv, WSP_NHI) << 32);
  sp.c_org = wave_get_u32(rv, WSP_CORG);
  sp.ea = (int)wave_get_u32(rv, WSP_EA);
  sp.eb = (int)wave_get_u32(rv, WSP_EB);
  sp.cum_a = wave_get_u32(rv, WSP_CUMA);
  const u32 lead = (u32)((u64)sp.dst & 127u);
  const u64 wofs = (u64)w * WGA_W_BYTES;
  u8* const B = sp.dst - lead + wofs;
  /* valid stage bytes [vlo, vhi) */
  const u32 vlo = w == 0u ? lead : 0u;
  const u64 end_rel = (u64)lead + sp.N - wofs; /* > 0 */
  const u32 vhi = end_rel < (u64)WGA_W_BYTES ? (u32)end_rel : WGA_W_BYTES;
  const u32 g_first = vlo >> 4, g_span = ((vhi - 1u) >> 4) - g_first;
  int cw0; /* column of stage byte 0 */
  u64 rare_shift = 0; /* a tail's windows are re-based: what the rare paths add to the piece's slice index */
  {
    const u64 lbase = wave_get_u64(rv, WSP_LBASE);
    const u32 lnum = wave_get_u32(rv, WSP_LNUM);
    if (!(wave_get_u32(rv, WSP_FLAGS) & WSF_TAIL)) {
      cw0 = (int)wave_get_u32(rv, WSP_C0) - (int)lead + (int)(u32)wofs;
      sp.lbuf = buf_make((const void*)lbase, lnum);
    } else {
      /* a tail has no events and may be longer than 2^31: every window is re-based on its own first byte, so that
       * all offsets inside it stay small (window byte t <-> slice offset t - lead from the shifted base) */
      cw0 = -(int)lead;
      sp.c_org = 0u;
      rare_shift = wofs;
      const u64 rem = end_rel < 0x40000000ull ? end_rel : 0x40000000ull;
      sp.lbuf = buf_make((const void*)(RC ? lbase - wofs : lbase + wofs), RC ? 0x80000020u : (u32)(rem + 48ull));
    }
  }

  /* ---- the events of this window: [e_lo, e_hi) start inside its valid columns.  After the search ONE batch of LDS
   * loads gives every lane all it will need of "its" event: lane l holds event e_lo - 1 + l (lane 0: the event in front
   * of the window, whose gap may reach into it) -------------------------------------------------------------------- */
  const int e_lo = ev_first_ge(sp.ev, sp.ea, sp.eb, cw0, lane);
  const int x_end = cw0 + (int)vhi;
  const int e_mine = e_lo - 1 + (int)lane;
  const int e_rd = e_mine < sp.ea ? sp.ea : (e_mine > sp.eb ? sp.eb : e_mine); /* a readable entry */
  EvView evw;
  ev_view(evw, sp.ev, e_rd);
  int e_hi;
  {
    const int k = (int)__popcll(__ballot((int)(lane != 0u) & (int)(e_mine < sp.eb) & (int)((int)evw.gs < x_end)));
    e_hi = (int)WGA_UNI32((u32)(e_lo + k));
    if (k == 63) /* more than 63 events in one window (indel-dense stretch): count on */
      for (;;) {
        const int p = e_hi + (int)lane;
        const int pc = p < sp.eb ? p : sp.eb;
        const int cv = (int)sp.ev.c(pc);
        const int k2 = (int)__popcll(__ballot((int)(p < sp.eb) & (int)(cv < x_end)));
        e_hi = (int)WGA_UNI32((u32)(e_hi + k2));
        if (k2 < 64) break;
      }
  }
  const u32 m_lo = wave_get_u32(evw.m0, 1); /* lane 1 holds entry e_lo (or eb: readable) */
  const u32 base_adj = m_lo - sp.cum_a;     /* gap bases of the piece's events in front of the window */
  /* a gap that starts in front of the window and reaches into it */
  const bool has_cover = e_lo > sp.ea && (int)(wave_get_u32(evw.gs, 0) + (m_lo - wave_get_u32(evw.m0, 0))) > cw0;

  /* ---- per-granule source adjustment: gap bases of the events that start before the granule -------------------- */
#pragma unroll
  for (u32 u = 0; u < WGA_W_U; u++) wt[u * 64u + lane] = 0u;
  if (lane == 0u) wt[WGA_W_G] = 0u;
  WGA_WAVE_SYNC();
  {
    const u32 gl = evw.m1 - evw.m0;
    const u32 g0 = (u32)((int)evw.gs - cw0) >> 4;
    if ((int)(lane != 0u) & (int)(e_mine < e_hi) & (int)(gl != 0u)) atomicAdd(&wt[g0 + 1u], gl);
  }
  for (int e0 = e_lo + 63; e0 < e_hi; e0 += 64) { /* events beyond the first 63 (rare) */
    const int e = e0 + (int)lane;
    const int ec = e < e_hi ? e : e_hi - 1;
    const u32 gl = sp.ev.m(ec + 1) - sp.ev.m(ec);
    const u32 g0 = (u32)((int)sp.ev.c(ec) - cw0) >> 4;
    if ((int)(e < e_hi) & (int)(gl != 0u)) atomicAdd(&wt[g0 + 1u], gl);
  }
  WGA_WAVE_SYNC();
  u32 adj[WGA_W_U];
  {
    u32 v[WGA_W_U], sum = 0;
#pragma unroll
    for (u32 u = 0; u < WGA_W_U; u++) {
      v[u] = wt[lane * WGA_W_U + u];
      sum += v[u];
    }
    u32 run = wave_incl_scan_u32(sum) - sum + base_adj;
    WGA_WAVE_SYNC();
#pragma unroll
    for (u32 u = 0; u < WGA_W_U; u++) {
      run += v[u];
      wt[lane * WGA_W_U + u] = run;
    }
    WGA_WAVE_SYNC();
#pragma unroll
    for (u32 u = 0; u < WGA_W_U; u++) adj[u] = wt[u * 64u + lane];
#if WGA_W_U == 4
    WGA_PIN4(adj[0], adj[1], adj[2], adj[3]); /* four loads, one wait, before the stage is written */
#endif
  }
  WGA_WAVE_SYNC(); /* the table is dead: the stage may be written */

  /* ---- plain fill (every granule as if no gap touched it) with the first round of pass 1 in flight next to it ------ */
  u32 badmask = 0u; /* bit u: fill granule u * 64 + lane, bit 31: the granule of pass 1 */
  const bool act0 = (bool)((int)(lane != 0u) & (int)(e_mine < e_hi)); /* this lane holds an event of the window */
  const bool cover0 = (bool)((int)(lane == 0u) & (int)has_cover);
  Fix1 f0;
  {
    u32 raw[WGA_W_U][4], loff[WGA_W_U];
    const int koff = cw0 - (int)sp.c_org;
#pragma unroll
    for (u32 u = 0; u < WGA_W_U; u++) {
      const u32 g = u * 64u + lane;
      u32 lo = w_loff<RC>(koff + (int)(g << 4) - (int)adj[u]);
      WGA_PIN(lo);
      loff[u] = (g - g_first <= g_span) ? lo : WGA_BUF_OOB; /* granules outside the piece: no load */
    }
#pragma unroll
    for (u32 u = 0; u < WGA_W_U; u++) buf_load16(sp.lbuf, loff[u], raw[u]);
    fix1_issue<RC>(f0, sp, evw, cw0, e_lo, e_rd, act0);
#pragma unroll
    for (u32 u = 0; u < WGA_W_U; u++) {
      u32 o[4], inv[4];
      w_finish<RC>(raw[u], o, inv);
      /* a candidate for InvalidBase: only windows the buffer returned whole (offset >= -16; predicated lanes and
       * granules inside a long gap got zeros, which are no bases) */
      if (RC) badmask |= ((int)(loff[u] <= 0x80000010u) & (int)((inv[0] | inv[1] | inv[2] | inv[3]) != 0u)) ? 1u << u : 0u;
      const u32x4_a16 ov = {o[0], o[1], o[2], o[3]};
      stage[u * 64u + lane] = ov;
    }
  }
  WGA_WAVE_SYNC();

  /* ---- fix-ups: one lane per event of the window (and the gap that reaches in from the front) ------------------- */
  if (fix1_finish<RC>(f0, sp, cw0, e_rd, stage, lowmask)) badmask |= 0x80000000u;
  const int g0_mine = f0.g0;
  WGA_WAVE_SYNC();
  fix23(evw, cw0, (bool)((int)act0 | (int)cover0), cover0, stage, lowmask, lane);
  WGA_WAVE_SYNC();
  for (int e0 = e_lo + 63; e0 < e_hi; e0 += 64) { /* more than 63 events in one window: indel-dense stretches */
    const int e = e0 + (int)lane;
    const bool act = e < e_hi;
    const int er = act ? e : e_lo;
    EvView ev2;
    ev_view(ev2, sp.ev, er);
    Fix1 f;
    fix1_issue<RC>(f, sp, ev2, cw0, e_lo, er, act);
    if (fix1_finish<RC>(f, sp, cw0, er, stage, lowmask)) {
      SpanW sr = sp;
      span_rare(sr, a, rec);
      sr.sbase += rare_shift;
      rescan_granule(sr, cw0, (u32)f.g0, vlo, vhi);
    }
    WGA_WAVE_SYNC();
    fix23(ev2, cw0, act, false, stage, lowmask, lane);
    WGA_WAVE_SYNC();
  }
  if (RC && __ballot(badmask != 0u)) { /* rare: find out exactly (utils.rs:97) */
    SpanW sr = sp;
    span_rare(sr, a, rec);
    sr.sbase += rare_shift;
#pragma unroll
    for (u32 u = 0; u < WGA_W_U; u++)
      if (badmask & (1u << u)) rescan_granule(sr, cw0, u * 64u + lane, vlo, vhi);
    if (badmask & 0x80000000u) rescan_granule(sr, cw0, (u32)g0_mine, vlo, vhi);
  }

  /* ---- copy-out: whole aligned lines; only the piece's first / last sixteen bytes can be partial ----------------- */
  const BufRsrc sbuf = buf_make(B, WGA_W_BYTES);
#pragma unroll
  for (u32 u = 0; u < WGA_W_U; u++) {
    const u32 g = u * 64u + lane, t = g << 4;
    const u32x4_a16 v = stage[g];
    const u32 o[4] = {v[0], v[1], v[2], v[3]};
    buf_store16_w(sbuf, ((int)(t >= vlo) & (int)(t + 16u <= vhi)) ? t : WGA_BUF_OOB, o);
  }
  if (((vlo | vhi) & 15u) != 0u && lane < 2u) { /* lane 0: the granule the piece starts in, lane 1: the one it ends in */
    const u32 gq = lane == 0u ? vlo >> 4 : vhi >> 4;
    u32 lo = gq << 4, hi = lo + 16u;
    lo = lo < vlo ? vlo : lo;
    hi = hi > vhi ? vhi : hi;
    if (lane == 1u && (vhi >> 4) == (vlo >> 4) && (vlo & 15u) != 0u) hi = lo; /* one granule holds both ends: lane 0 has it */
    if ((lo & 15u) == 0u && hi == lo + 16u) hi = lo;                           /* a whole granule: stored above          */
    const u8* const sb = (const u8*)stage;
#pragma clang loop vectorize(disable) unroll(disable)
    for (u32 j = lo; j < hi; j++) B[j] = sb[j]; /* byte stores, never read-modify-write */
  }
  WGA_WAVE_SYNC(); /* the stage is rewritten by this wave's next window */
}

/* A row piece byte by byte, by one wave: a slice at a pool edge (its window loads would need bounds checks) or a record
 * whose slices are shorter than its CIGAR consumes (with the check for String::insert_str beyond the end, cigar.rs:507,513).
 * Correctness only: no consistent PAF whose sequences sit inside the pool gets here. */
__device__ __forceinline__ void emit_span_bytes(KArgP a, const u32* rec, const u32* s_col, const u32* s_cum,
                                                u64 tile_start, u32 lane) {
  SpanW sp;
  const u32 flags = rec[WSP_FLAGS], r = rec[WSP_REC];
  sp.ev.col = s_col + rec[WSP_EVBASE];
  sp.ev.cum = s_cum + rec[WSP_EVBASE];
  sp.dst = (u8*)((u64)rec[WSP_DST] | ((u64)rec[WSP_DST + 1] << 32));
  sp.N = (u64)rec[WSP_NLO] | ((u64)rec[WSP_NHI] << 32);
  sp.c0 = (int)rec[WSP_C0];
  sp.c_org = rec[WSP_CORG];
  sp.ea = (int)rec[WSP_EA];
  sp.eb = (int)rec[WSP_EB];
  sp.cum_a = rec[WSP_CUMA];
  span_rare(sp, a, rec);
  const bool is_q = (flags & WSF_Q) != 0u;
  const wga_rec_desc* rp = a->recs + r;
  RowSrc src;
  src.fa = is_q ? a->q_fa : a->t_fa;
  src.fa_bytes = is_q ? a->q_fa_bytes : a->t_fa_bytes;
  src.src_off = is_q ? rp->q_src_off : rp->t_src_off;
  src.src_len = sp.src_len;
  src.rc = (flags & WSF_RC) != 0u;
  if (flags & WSF_PANIC) { /* an I (D) op whose target (query) consumption so far exceeds the fetched slice */
    bool pan = false;
    for (int i = sp.ea + (int)lane; i < sp.eb; i += 64)
      pan |= sp.sbase + (u64)(sp.ev.c(i) - sp.c_org) - (u64)(sp.ev.m(i) - sp.cum_a) > sp.src_len;
    if (pan) { /* the exact op: serial walk of the segment by the detecting lane */
      const u64 rs = a->op_off[r];
      u64 pos = sp.sbase;
      for (u64 k = tile_start + rec[WSP_KA]; k < tile_start + rec[WSP_KB]; k++) {
        const u32 op = a->ops[k];
        const u32 c = op_class(op & 15u);
        const u64 len = op >> 4;
        if (c == (is_q ? CLS_D : CLS_I) && pos > sp.src_len) {
          atomicMin((u64*)&a->diag[r].panic_op_idx, k - rs);
          break;
        }
        if (c == CLS_MX || c == (is_q ? CLS_I : CLS_D)) pos += len;
      }
    }
  }
#pragma clang loop vectorize(disable) unroll(disable)
  for (u64 x = lane; x < sp.N; x += 64u) {
    const u32 c = (u32)(sp.c0 + (int)(u32)x);
    /* lo = events of the piece that start at or before c */
    int lo = sp.ea, hi = sp.eb;
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (sp.ev.c(mid) <= c)
        lo = mid + 1;
      else
        hi = mid;
    }
    u8 o;
    bool dash = false;
    /* the last event with a gap that starts at or before c decides (zero-length events in between do not matter) */
    int i = lo - 1;
    while (i >= sp.ea && sp.ev.m(i + 1) == sp.ev.m(i)) i--;
    if (i >= sp.ea) dash = c - sp.ev.c(i) < sp.ev.m(i + 1) - sp.ev.m(i);
    if (dash) {
      o = (u8)'-';
    } else {
      const u64 sidx = (flags & WSF_TAIL) ? sp.sbase + x : sp.sbase + (u64)(c - sp.c_org) - (u64)(sp.ev.m(lo) - sp.cum_a);
      o = src_byte(src, sidx, sp.bad_base_pos);
    }
    sp.dst[x] = o;
  }
}

/* (column, I | D << 16 gap-op counts) in front of tile-relative op k, from the per-thread prefixes of phase A */
__device__ __forceinline__ void prefix_at(KArgP a, u64 tile_start, u32 nt, u32 k, u32 tot_col, u32 tot_cnt,
                                          const u32 (*s_pref)[2], u32& col, u32& cnt) {
  col = tot_col;
  cnt = tot_cnt;
  if (k < nt) {
    col = s_pref[k >> 2][0];
    cnt = s_pref[k >> 2][1];
    for (u32 e = 0; e < (k & 3u); e++) {
      const u32 op = a->ops[tile_start + (k & ~3u) + e];
      const u32 cl = op_class(op & 15u);
      col += cl <= CLS_D ? (op >> 4) : 0u;
      cnt += cl == CLS_I ? 1u : (cl == CLS_D ? 0x10000u : 0u);
    }
  }
}

/* One planning round, by ONE wave and in vector code: lane = (segment, job), 16 consecutive record segments of the tile x
 * (target row, query row, target tail, query tail).  Every lane works out its piece from the record descriptor, the op
 * offsets and the phase-A prefixes — no scalar state, nothing that outlives the round — and the active lanes write their
 * records, in order, to s_span.  s_plan = {pieces, windows, more segments follow}. */
__device__ __forceinline__ void plan_round(KArgP a, const u64 g, const u32 pre, u32 seg_base, u64 tile_start,
                                           u64 tile_end, u32 tot_col, u32 tot_cnt, const u32 (*s_pref)[2], const u32* s_cum,
                                           u32* s_span, u32* s_plan, u32 lane) {
  const u32 nt = (u32)(tile_end - tile_start);
  const u32 r0 = wave_get_u32(pre, 2);
  const u32 sl = lane >> 2, job = lane & 3u;
  const bool is_q = (job & 1u) != 0u, is_tail = job >= 2u;
  const u64 r = (u64)r0 + seg_base + sl;
  /* The tile's first record is described by the tile descriptor (its op range and a copy of its record descriptor sit in
   * the lanes of `pre`): the common tile — one record segment — is planned without a single dependent load.  Later
   * records load their op range, and only if the first one ends inside the tile. */
  const bool first = seg_base + sl == 0u;
  const u64 re0 = wave_get_u64(pre, 12);
  const bool valid = first || (re0 < tile_end && r < (u64)a->n_rec);
  u64 e_a = wave_get_u64(pre, 10), e_b = re0;
  if (!first) {
    e_a = e_b = 0xFFFFFFFFFFFFFFFFull;
    if (valid) {
      e_a = a->op_off[r];
      e_b = a->op_off[r + 1];
    }
  }
  const u64 lo_op = e_a > tile_start ? e_a : tile_start, hi_op = e_b < tile_end ? e_b : tile_end;
  const bool nonempty = valid && lo_op < hi_op;
  const u64 m_more = __ballot(valid && e_b < tile_end);
  const u64 p_mx = wave_get_u64(pre, 4), p_i = wave_get_u64(pre, 6), p_d = wave_get_u64(pre, 8);
  /* the first record's descriptor (wga_tile_desc dwords 14..31, flags in dword 3) */
  const u32 d_flags = wave_get_u32(pre, 3);
  const u64 d_trow = wave_get_u64(pre, 14), d_qrow = wave_get_u64(pre, 16), d_toff = wave_get_u64(pre, 18), d_tlen = wave_get_u64(pre, 20),
            d_qoff = wave_get_u64(pre, 22), d_qlen = wave_get_u64(pre, 24), d_I = wave_get_u64(pre, 26), d_D = wave_get_u64(pre, 28),
            d_L = wave_get_u64(pre, 30);
  u32 N_lo = 0u, N_hi = 0u, nwin = 0u;
  u32 recw[WGA_W_SPAN_WORDS];
#pragma unroll
  for (u32 k = 0; k < WGA_W_SPAN_WORDS; k++) recw[k] = 0u;
  bool active = false;
  if (nonempty) {
    const u32 ka = (u32)(lo_op - tile_start), kb = (u32)(hi_op - tile_start);
    u32 col_a, cnt_a, col_b, cnt_b;
    prefix_at(a, tile_start, nt, ka, tot_col, tot_cnt, s_pref, col_a, cnt_a);
    prefix_at(a, tile_start, nt, kb, tot_col, tot_cnt, s_pref, col_b, cnt_b);
    u32 rflags = d_flags;
    u64 row_off = is_q ? d_qrow : d_trow, src_off = is_q ? d_qoff : d_toff, src_len = is_q ? d_qlen : d_tlen;
    u64 gap_total = is_q ? d_D : d_I, L = d_L;
    if (!first) {
      const wga_rec_desc* rp = a->recs + r;
      rflags = (u32)rp->neg;
      row_off = is_q ? rp->q_row_off : rp->t_row_off;
      src_off = is_q ? rp->q_src_off : rp->t_src_off;
      src_len = is_q ? rp->q_src_len : rp->t_src_len;
      gap_total = is_q ? rp->D_total : rp->I_total;
      L = rp->L;
    }
    /* only the tile's first record can continue from earlier tiles: its sums are in the tile descriptor */
    const bool cont = e_a < tile_start;
    const u64 b_mx = cont ? p_mx : 0ull, b_i = cont ? p_i : 0ull, b_d = cont ? p_d : 0ull;
    const u64 cb = b_mx + b_i + b_d;
    const u64 row_len = src_len + gap_total;
    const u32 seg_cols = col_b - col_a;
    const bool panic = (rflags & WGA_K2W_PANIC_FLAG) != 0u;
    u64 x0, nbytes;
    if (!is_tail) {
      const u64 x1 = cb + seg_cols < row_len ? cb + seg_cols : row_len;
      x0 = cb;
      nbytes = x1 > cb ? x1 - cb : 0;
      active = nbytes != 0 || panic; /* a record that may panic is checked even where its rows are cut off */
    } else {
      x0 = L;
      nbytes = (e_b <= tile_end && (rflags & (is_q ? 4u : 2u)) && row_len > L) ? row_len - L : 0;
      active = nbytes != 0;
    }
    if (active) {
      const u64 sbase = is_tail ? L - gap_total : (is_q ? b_mx + b_i : b_mx + b_d);
      const bool rc = is_q && (rflags & 1u) != 0u;
      const u8* const fa = is_q ? a->q_fa : a->t_fa;
      const u64 fa_bytes = is_q ? a->q_fa_bytes : a->t_fa_bytes;
      const bool safe = src_off >= 16 && src_off + src_len + 16 <= fa_bytes;
      const bool bytes = !safe || panic;
      const u64 dst = (u64)(a->out + row_off + x0);
      const u32 evbase = is_q ? (tot_cnt & 0xFFFFu) + 2u : 0u;
      const u32 ea = is_tail ? (tot_cnt & 0xFFFFu) : (is_q ? cnt_a >> 16 : cnt_a & 0xFFFFu);
      const u32 eb = is_tail ? ea : (is_q ? cnt_b >> 16 : cnt_b & 0xFFFFu);
      /* Source windows as a range-checked buffer around the slice: the plain fill also computes offsets for granules
       * that lie inside a long gap (column minus ALL the gap's bases: far in front of the slice) — those loads must
       * return nothing instead of touching memory.  Forward rows: base = slice index sbase - 16, offsets up to the
       * rest of the slice + 16 are in range (the slice sits >= 16 bytes inside the pool); reversed rows walk down
       * from base + 2^31 = the mirrored window of sbase: only offsets >= -16 are in range, the low side is bounded
       * by the piece's own valid range. */
      const u64 win_base = (u64)(rc ? fa + src_off + src_len - 16 - sbase : fa + src_off + sbase);
      const u64 lbase = win_base - (rc ? 0x80000000ull : 16ull);
      const u64 rem = src_len > sbase ? src_len - sbase : 0ull;
      const u32 lnum = rc ? 0x80000020u : (u32)(rem + 32ull < 0x80000000ull ? rem + 32ull : 0x80000000ull);
      N_lo = (u32)nbytes;
      N_hi = (u32)(nbytes >> 32);
      const u64 nw64 = (((u64)(dst & 127u)) + nbytes + WGA_W_BYTES - 1u) / WGA_W_BYTES;
      nwin = bytes ? 1u : (nw64 < 0x7FFFFFFFull ? (u32)nw64 : 0x7FFFFFFFu);
      recw[WSP_DST] = (u32)dst;
      recw[WSP_DST + 1] = (u32)(dst >> 32);
      recw[WSP_NLO] = N_lo;
      recw[WSP_NHI] = N_hi;
      recw[WSP_C0] = is_tail ? 0u : col_a;
      recw[WSP_CORG] = is_tail ? 0u : col_a;
      recw[WSP_EA] = ea;
      recw[WSP_EB] = eb;
      recw[WSP_CUMA] = s_cum[(is_tail ? 0u : evbase) + ea];
      recw[WSP_EVBASE] = is_tail ? 0u : evbase;
      recw[WSP_LBASE] = (u32)lbase;
      recw[WSP_LBASE + 1] = (u32)(lbase >> 32);
      recw[WSP_LNUM] = lnum;
      recw[WSP_FLAGS] = (rc ? WSF_RC : 0u) | (bytes ? WSF_BYTES : 0u) | (is_q ? WSF_Q : 0u) | (is_tail ? WSF_TAIL : 0u) |
                        ((panic && !is_tail) ? WSF_PANIC : 0u);
      recw[WSP_REC] = (u32)r;
      recw[WSP_KA] = ka;
      recw[WSP_KB] = kb;
      recw[WSP_SBASE] = (u32)sbase;
      recw[WSP_SBASE + 1] = (u32)(sbase >> 32);
    }
  }
  /* the active lanes' records go to the table in lane order = (segment, job) order */
  const u64 m_act = __ballot(active);
  const u32 idx = lane_rank(m_act, lane);
  const u32 w_incl = wave_incl_scan_u32(nwin);
  if (active) {
    recw[WSP_W0] = w_incl - nwin;
    u32* const d = s_span + idx * WGA_W_SPAN_WORDS;
#pragma unroll
    for (u32 k = 0; k < WGA_W_SPAN_WORDS; k++) d[k] = recw[k];
  }
  if (lane == 63u) {
    s_plan[0] = (u32)__popcll(m_act);
    s_plan[1] = w_incl;
    s_plan[2] = ((m_more >> 60) & 1ull) ? 1u : 0u; /* the round's last record ends inside the tile: more segments follow */
  }
}

/* The plan of the COMMON tile, worked out ahead of the row kernel (one thread per tile, next to k_tile_base): when the
 * tile's first record covers the whole tile — one record segment, no slice tail, no byte-wise row — its two pieces
 * (target row, query row) are known up to what phase A counts (the tile's columns and gap ops).  The row kernel loads the
 * two records with the tile descriptor, patches those counts in and starts on the windows: no planner wave, no second
 * barrier.  Bit 31 of the first record's flags says that the tile is of this kind. */
#define WSF_SIMPLE 0x80000000u
#define WGA_W_PLAN_WORDS (2u * WGA_W_SPAN_WORDS)
__global__ __launch_bounds__(256) void k_tile_plan(const wga_tile_desc* __restrict__ descs, u64 n_ops, u8* out,
                                                   const u8* t_fa, u64 t_fa_bytes, const u8* q_fa, u64 q_fa_bytes,
                                                   u32* __restrict__ plan) {
  const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 tile_start = g * WGA_TILE;
  if (tile_start >= n_ops) return;
  const u64 tile_end = tile_start + WGA_TILE < n_ops ? tile_start + WGA_TILE : n_ops;
  const wga_tile_desc d = descs[g];
  u32* const p = plan + g * WGA_W_PLAN_WORDS;
  const u32 rflags = d.neg;
  const bool cont = d.rs < tile_start;
  const u64 b_mx = cont ? d.b_mx : 0ull, b_i = cont ? d.b_i : 0ull, b_d = cont ? d.b_d : 0ull;
  const u64 cb = b_mx + b_i + b_d;
  bool simple = d.re >= tile_end && d.tile_cols <= WGA_FAST_COL_LIMIT && !(rflags & WGA_K2W_PANIC_FLAG) &&
                !(d.re == tile_end && (rflags & 6u));
  u32 w[WGA_W_PLAN_WORDS];
  for (u32 k = 0; k < WGA_W_PLAN_WORDS; k++) w[k] = 0u;
  for (u32 q = 0; q < 2u && simple; q++) {
    const bool is_q = q != 0u;
    const u64 row_off = is_q ? d.q_row_off : d.t_row_off, src_off = is_q ? d.q_src_off : d.t_src_off;
    const u64 src_len = is_q ? d.q_src_len : d.t_src_len, gap_total = is_q ? d.D_total : d.I_total;
    const u64 fa_bytes = is_q ? q_fa_bytes : t_fa_bytes;
    const u8* const fa = is_q ? q_fa : t_fa;
    if (!(src_off >= 16 && src_off + src_len + 16 <= fa_bytes)) { /* a slice at a pool edge: byte by byte, planned in the kernel */
      simple = false;
      break;
    }
    const u64 row_len = src_len + gap_total;
    const u64 xmax = row_len > cb ? row_len - cb : 0ull;
    const u64 sbase = is_q ? b_mx + b_i : b_mx + b_d;
    const bool rc = is_q && (rflags & 1u) != 0u;
    const u64 dst = (u64)(out + row_off + cb);
    const u64 win_base = (u64)(rc ? fa + src_off + src_len - 16 - sbase : fa + src_off + sbase);
    const u64 lbase = win_base - (rc ? 0x80000000ull : 16ull);
    const u64 rem = src_len > sbase ? src_len - sbase : 0ull;
    u32* const r = w + q * WGA_W_SPAN_WORDS;
    r[WSP_DST] = (u32)dst;
    r[WSP_DST + 1] = (u32)(dst >> 32);
    r[WSP_NLO] = xmax < 0x7FFFFFFFull ? (u32)xmax : 0x7FFFFFFFu; /* the kernel takes min(tile columns, this) */
    r[WSP_LBASE] = (u32)lbase;
    r[WSP_LBASE + 1] = (u32)(lbase >> 32);
    r[WSP_LNUM] = rc ? 0x80000020u : (u32)(rem + 32ull < 0x80000000ull ? rem + 32ull : 0x80000000ull);
    r[WSP_FLAGS] = (rc ? WSF_RC : 0u) | (is_q ? WSF_Q : 0u);
    r[WSP_REC] = d.rec;
    r[WSP_KB] = (u32)(tile_end - tile_start);
    r[WSP_SBASE] = (u32)sbase;
    r[WSP_SBASE + 1] = (u32)(sbase >> 32);
  }
  if (simple) w[WSP_FLAGS] |= WSF_SIMPLE;
  for (u32 k = 0; k < WGA_W_PLAN_WORDS; k++) p[k] = simple ? w[k] : 0u;
}

__device__ __forceinline__ void expand_tile_w(const ExpandArgs& a, const u64 g) {
  __shared__ u32 s_col[WGA_TILE + 8];  /* gap events: start column; target row from entry 0, then the query row                */
  __shared__ u32 s_cum[WGA_TILE + 8];  /*             gap bases of the row's events before                                 */
  __shared__ u32 s_pref[WGA_BLOCK][2]; /* (column, I | D << 16 gap-op counts) before every thread's first op              */
  __shared__ u32 s_tot[2];             /* ... and at the end of the tile                                                   */
  __shared__ u32 s_w4[16];
  __shared__ u32 s_plan[4];
  __shared__ u32x4_a16 s_lowmask[17];
  __shared__ u32x4_a16 s_stage[4][WGA_W_G];
  __shared__ u32 s_span[WGA_W_MAXSPAN * WGA_W_SPAN_WORDS];

  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  const u64 tile_start = g * WGA_TILE;
  const u64 tile_end = tile_start + WGA_TILE < a.n_ops ? tile_start + WGA_TILE : a.n_ops;
  const u32 nt = (u32)(tile_end - tile_start);
  build_lowmask(s_lowmask);

  u32 pre = 0u, planv = 0u;
  if (lane < 32u) pre = ((const u32*)(a.tdesc + g))[lane];
  if (lane < WGA_W_PLAN_WORDS) planv = a.plan[g * WGA_W_PLAN_WORDS + lane]; /* the two pieces of a one-segment tile (k_tile_plan) */
  const u64 tile_cols = wave_get_u64(pre, 0);
  if (a.force_slow || tile_cols > WGA_FAST_COL_LIMIT) return; /* listed for the op-serial walk (block-uniform) */

  /* ---- phase A: 4 consecutive ops per thread, block scan, compact gap lists ---------------------------------------- */
  u32 opw[4];
  {
    const u32 base = tid * 4u;
    if (base + 3 < nt) {
      const u32x4_a16 v = *(const u32x4_a16*)(a.ops + tile_start + base);
      opw[0] = v[0], opw[1] = v[1], opw[2] = v[2], opw[3] = v[3];
    } else {
      for (int e = 0; e < 4; e++) opw[e] = (base + e < nt) ? a.ops[tile_start + base + e] : 0u;
    }
  }
  {
    u32 cls[4], l[4], sl = 0, si = 0, sd = 0, cnt = 0;
    for (int e = 0; e < 4; e++) {
      const u32 code = opw[e] & 15u, len = opw[e] >> 4;
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
    const u32 q_base = (stot[3] & 0xFFFFu) + 2u; /* the query-row list starts behind the target-row list and its sentinels */
    s_pref[tid][0] = x_col;
    s_pref[tid][1] = x_cnt;
    for (int e = 0; e < 4; e++) {
      const bool isi = cls[e] == CLS_I, isd = cls[e] == CLS_D;
      if (isi | isd) {
        const u32 len = opw[e] >> 4;
        const u32 slot = isi ? (x_cnt & 0xFFFFu) : q_base + (x_cnt >> 16);
        s_col[slot] = x_col;
        s_cum[slot] = isi ? x_i : x_d;
        x_i += isi ? len : 0u;
        x_d += isi ? 0u : len;
        x_cnt += isi ? 1u : 0x10000u;
      }
      x_col += l[e];
    }
    if (tid == WGA_BLOCK - 1) { /* totals + two sentinels per list */
      s_tot[0] = x_col;
      s_tot[1] = x_cnt;
      const u32 ni = x_cnt & 0xFFFFu, nd = x_cnt >> 16;
      s_col[ni] = s_col[ni + 1u] = x_col;
      s_cum[ni] = s_cum[ni + 1u] = x_i;
      s_col[ni + 2u + nd] = s_col[ni + 3u + nd] = x_col;
      s_cum[ni + 2u + nd] = s_cum[ni + 3u + nd] = x_d;
    }
  }
  __syncthreads();

  /* ---- phase B: rounds of 16 record segments: wave 0 plans the pieces, the four waves take their windows round robin --- */
  u32x4_a16* const stage = s_stage[wave];
  const u32 tot_col = WGA_UNI32(s_tot[0]), tot_cnt = WGA_UNI32(s_tot[1]);
  KArgP ka = WGA_KARG_PTR(a);
  const bool simple = (wave_get_u32(planv, WSP_FLAGS) & WSF_SIMPLE) != 0u; /* block-uniform */
  for (u32 seg_base = 0;; seg_base += WGA_W_SEGS) {
    u32 nspan, nwin_all, more;
    if (simple) {
      /* every wave patches what phase A counted into the two prepared records and writes them itself (all four write
       * the same words): no planner, no barrier */
      const u32 n_t = wave_get_u32(planv, WSP_NLO), n_q = wave_get_u32(planv, WGA_W_SPAN_WORDS + WSP_NLO);
      const u32 N_t = n_t < tot_col ? n_t : tot_col, N_q = n_q < tot_col ? n_q : tot_col;
      const u32 lead_t = wave_get_u32(planv, WSP_DST) & 127u, lead_q = wave_get_u32(planv, WGA_W_SPAN_WORDS + WSP_DST) & 127u;
      const u32 nw_t = N_t ? (lead_t + N_t + WGA_W_BYTES - 1u) / WGA_W_BYTES : 0u;
      const u32 nw_q = N_q ? (lead_q + N_q + WGA_W_BYTES - 1u) / WGA_W_BYTES : 0u;
      u32 v = planv;
      v = lane == WSP_NLO ? N_t : v;
      v = lane == WGA_W_SPAN_WORDS + WSP_NLO ? N_q : v;
      v = lane == WSP_EB ? (tot_cnt & 0xFFFFu) : v;
      v = lane == WGA_W_SPAN_WORDS + WSP_EB ? (tot_cnt >> 16) : v;
      v = lane == WGA_W_SPAN_WORDS + WSP_EVBASE ? (tot_cnt & 0xFFFFu) + 2u : v;
      v = lane == WGA_W_SPAN_WORDS + WSP_W0 ? nw_t : v;
      v = lane == WSP_FLAGS ? (v & ~WSF_SIMPLE) : v;
      if (lane < WGA_W_PLAN_WORDS) s_span[lane] = v;
      WGA_WAVE_SYNC();
      nspan = 2u;
      nwin_all = nw_t + nw_q;
      more = 0u;
    } else {
      WGA_KARG_FRESH(ka);
      if (wave == 0u) plan_round(ka, g, pre, seg_base, tile_start, tile_end, tot_col, tot_cnt, s_pref, s_cum, s_span, s_plan, lane);
      __syncthreads();
      nspan = WGA_UNI32(s_plan[0]);
      nwin_all = WGA_UNI32(s_plan[1]);
      more = WGA_UNI32(s_plan[2]);
    }
    u32 k = 0u; /* the piece that holds window wi */
    for (u32 wi = wave; wi < nwin_all; wi += 4u) {
      while (k + 1u < nspan && WGA_UNI32(s_span[(k + 1u) * WGA_W_SPAN_WORDS + WSP_W0]) <= wi) k++;
      const u32* const rec = s_span + k * WGA_W_SPAN_WORDS;
      const u32 flags = WGA_UNI32(rec[WSP_FLAGS]);
      const u32 w = wi - WGA_UNI32(rec[WSP_W0]);
      if (flags & WSF_BYTES)
        emit_span_bytes(ka, rec, s_col, s_cum, tile_start, lane);
      else if (flags & WSF_RC)
        emit_window<true>(ka, rec, w, s_col, s_cum, stage, s_lowmask, lane);
      else
        emit_window<false>(ka, rec, w, s_col, s_cum, stage, s_lowmask, lane);
    }
    if (!more) break;
    __syncthreads(); /* the table is rewritten by the next round */
  }
}

__global__ __launch_bounds__(256, WGA_K2W_BLOCKS) void k_paf2maf_expand_w(ExpandArgs a) {
  expand_tile_w(a, xcd_tile_of_block());
}

/* tiles the window kernel leaves to v1's op-serial walk: beyond 2^31 columns, or all of them under "expand_force_slow" */
__global__ __launch_bounds__(256) void k_list_slow_tiles(const wga_tile_desc* descs, u64 nt, int all, u32* count, u32* list) {
  const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
  if (g >= nt) return;
  if (all || descs[g].tile_cols > WGA_FAST_COL_LIMIT) list[atomicAdd(count, 1u)] = (u32)g;
}

#endif
