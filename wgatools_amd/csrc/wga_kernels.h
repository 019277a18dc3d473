/*
 * wga_kernels.h — hand-written HIP kernels for gfx950 (CDNA4, wave64) of the wgatools CIGAR hot
 * path.  Integer / byte work, HBM-bound: no MFMA anywhere.  Included by wga_capi.cpp.
 *
 * Work decomposition (all op-stream kernels): the packed op stream of a whole batch is cut into
 * *globally aligned tiles* of WGA_TILE ops, independent of record boundaries, so the load is
 * balanced whatever the record-length skew (1-op records and 2 Mop records in one launch).  Small
 * pre-pass kernels (k_tile_rec, k_rec_desc, k_tile_base) tell every tile which record its first op
 * belongs to and where that record stands, so that the walk kernels start with one load.
 *
 *   K1 k_cigar_stat      one wave per tile; 16 B/lane coalesced op loads; per-segment wave
 *                        reduction; writes per-record counts (atomics only for records that span
 *                        tiles) and an 80-byte tile summary (class sums of the tile and of its
 *                        last segment) that lets any later kernel place a tile inside a long
 *                        record by summing summaries instead of rescanning ops.
 *   K2 k_paf2maf_expand  one 256-thread block per tile; block scan of the tile's ops into LDS
 *                        (compacted per-row gap lists + a per-16-column granule table); every
 *                        lane then owns 16-column output granules: a table lookup says "plain
 *                        copy" (one byte-unaligned 16 B window load, reverse-complement fused for
 *                        '-' strand), "all dashes", or "touches a gap" (queued and assembled from
 *                        two windows under byte masks).  Raw buffer loads / stores with
 *                        out-of-range offsets as the lane predicate; no read-modify-write.
 *   (K3..K8 are in wga_kernels2.h.)
 *
 * The same source also compiles under tests/emu/simt_emu.h for CPU-side logic tests: wga_intrin.h is the one place that
 * knows (it hands its names over to tests/emu/wga_intrin_emu.h there).
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
  u64 rec; /* record that owns the first op of the tile (saves later kernels the search) */
};

/* a 16-byte vector that only promises dword alignment (global_load_dwordx4 needs no more) */
typedef u32 u32x4_a4 __attribute__((vector_size(16), aligned(4)));
typedef u32 u32x4_a16 __attribute__((vector_size(16), aligned(16)));
/* byte-aligned 16-byte access: one global_load/store_dwordx4 on gfx950 (unaligned loads measure
 * the same as aligned ones, unaligned stores ~10 % slower: scripts/micro/unaligned_copy.hip) */
typedef u32 u32x4_a1 __attribute__((vector_size(16), aligned(1)));

#include <type_traits>

#include "wga_intrin.h" /* the gfx950 instructions the kernels are written with (cross-lane, buffer, uniformity) */

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}


#ifndef WGA_K1_BLOCKS
#define WGA_K1_BLOCKS 5 /* blocks per CU the register budget of k_cigar_stat is sized for: 94 VGPRs, no scratch.  6 (80 VGPRs + 12 B of scratch) measured 0.56 ms on one box and 0.70 ms on two others against 0.54-0.57 ms for 5 */
#endif
#ifndef WGA_K1_WHOLE_TILE_PATH
#define WGA_K1_WHOLE_TILE_PATH 0 /* 1: a second copy of the class-sum loop without the range test for segments that are the whole tile (3 of 24 instructions per op less, 8 registers spilled) */
#endif
#ifndef WGA_K1_LANE_STORE
#define WGA_K1_LANE_STORE 1 /* K1 writes its counters one field per lane (v_writelane) instead of from lane 0 */
#endif



/* exact sum of 64 u32 values (up to 2^38): two 16-bit halves scanned separately */
__device__ __forceinline__ u64 wave_sum_u32_wide(u32 v) {
  const u32 lo = wave_last_u32(wave_incl_scan_u32(v & 0xFFFFu));
  const u32 hi = wave_last_u32(wave_incl_scan_u32(v >> 16));
  return ((u64)hi << 16) + (u64)lo;
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v) { return wave_last_u32(wave_incl_scan_u32(v)); }
/* inclusive scan of a u64 over the 64 lanes */
__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v, u32 lane) {
#pragma unroll
  for (u32 d = 1; d < 64; d <<= 1) {
    const u64 o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
  for (int m = 32; m >= 1; m >>= 1) {
    u32 o = __shfl_xor(v, m);
    v = o < v ? o : v;
  }
  return v;
}

/* exclusive scan of four u32 values across a 256-thread block; tot[] = block totals.
 * s_w4: 16 words of LDS. */
__device__ __forceinline__ void block_excl_scan4_u32(const u32 v[4], u32 ex[4], u32 tot[4],
                                                     u32* s_w4, bool s_w4_idle = false) {
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  u32 inc[4];
#pragma unroll
  for (int k = 0; k < 4; k++) inc[k] = wave_incl_scan_u32(v[k]);
  if (!s_w4_idle) __syncthreads(); /* protect s_w4 reuse */
  if (lane == 63u) {
#pragma unroll
    for (int k = 0; k < 4; k++) s_w4[wave * 4 + k] = inc[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    u32 pre = 0, t = 0;
#pragma unroll
    for (u32 w = 0; w < 4; w++) {
      const u32 x = s_w4[w * 4 + k];
      pre += w < wave ? x : 0u;
      t += x;
    }
    ex[k] = pre + inc[k] - v[k];
    tot[k] = t;
  }
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
/* record that holds the first op of every tile: one thread per tile, plain binary search.  Done
 * ahead of the walk kernels so that their waves start with one load instead of a chain of
 * dependent probes. */
struct wga_tile_rec {
  u32 rec, neg; /* the record and its strand */
  u64 rs, re;   /* op_off[rec], op_off[rec + 1] */
};
__global__ __launch_bounds__(256) void k_tile_rec(const u64* __restrict__ op_off,
                                                  const u8* __restrict__ strand_neg, u32 n, u64 n_ops,
                                                  wga_tile_rec* __restrict__ tile_rec) {
  const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 x = g * WGA_TILE;
  if (x >= n_ops) return;
  u32 lo = 0, hi = n; /* last r with op_off[r] <= x; op_off[0] == 0, op_off[n] == n_ops > x */
  while (hi - lo > 1u) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (op_off[mid] <= x)
      lo = mid;
    else
      hi = mid;
  }
  wga_tile_rec t;
  t.rec = lo;
  t.neg = strand_neg[lo] != 0 ? 1u : 0u;
  t.rs = op_off[lo];
  t.re = op_off[lo + 1];
  tile_rec[g] = t;
}

/* one tile of the stat walk; w = the tile's packed ops, 16 per lane: op (j*64+lane)*4+e */
__device__ __forceinline__ void cigar_stat_tile(const u64 g, const u32 (&w)[16], const u32 lane, const u64* __restrict__ op_off,
                                                const u8* __restrict__ strand_neg, u64 n_ops,
                                                const wga_tile_rec* __restrict__ tile_rec, wga_cigar_counts* counts,
                                                wga_rec_diag* diag, wga_tile_sum* tiles) {
  const u64 tile_start = g * WGA_TILE;
  const u64 tile_end = tile_start + WGA_TILE < n_ops ? tile_start + WGA_TILE : n_ops;

  /* the first segment's record, bounds and strand arrive with the ops (k_tile_rec); later
   * segments — records that start inside the tile — load theirs */
  const wga_tile_rec tr = tile_rec[g];
  u32 r = WGA_UNI32(tr.rec);
  const u32 r_first = r;
  u64 cur = tile_start;
  u64 tot[5] = {0, 0, 0, 0, 0}, tail[5] = {0, 0, 0, 0, 0};
  u64 re = WGA_UNI64(tr.re);
  /* a class sum is `len & mask`, the mask one v_bfe_i32 of a class constant by the packed op itself (bit c set: code c belongs
   * to the class; the 16 bits stand twice, so that the length's lowest bit — bit 4 of the op — picks either copy) — 17 vector
   * instructions per op (round 5: 21 with the code cut out first and the D bases summed on their own; compares and selects on
   * a class number took 45), 20 with the range test of a segment that is not the whole tile */
  constexpr u32 MX_BITS = 0x01810181u, I_BITS = 0x02020202u, X_BITS = 0x01000100u, RARE_BITS = 0xF878F878u;
  constexpr u32 IEV_BITS = 0x00020002u, DEV_BITS = 0x00040004u; /* an I (not the rest of a split one), a D */
  const u32 lane4 = lane * 4u;
  /* the wave's sums: one 32-bit reduction each when no lane's lengths add up to 2^26 (64 lanes stay below 2^32: every tile of a
   * real alignment), else two 16-bit halves each */
  auto wave_sums = [&](u32 s_mx, u32 s_i, u32 s_t, u32 s_x, u32 ev, u64& Smx, u64& Si, u64& St, u64& Sx, u32& EV) {
    if (__ballot(s_t >= (1u << 26)) == 0ull) { /* wave-uniform */
      Smx = wave_sum_u32(s_mx), Si = wave_sum_u32(s_i), St = wave_sum_u32(s_t), Sx = wave_sum_u32(s_x);
    } else {
      Smx = wave_sum_u32_wide(s_mx), Si = wave_sum_u32_wide(s_i), St = wave_sum_u32_wide(s_t), Sx = wave_sum_u32_wide(s_x);
    }
    EV = wave_sum_u32(ev);
  };
  /* The WHOLE tile first, without a range test: a tile's last segment (its only one, in four tiles of five on 5-kop records) is
   * what the segments in front leave of these sums, so only segments that end inside the tile pay the three instructions per op
   * of the test — unless the tile holds an op outside M = X I D (an error for the run: every segment is then measured on its own,
   * with its first bad op). */
  u64 Wmx, Wi, Wt, Wx;
  u32 Wev;
  bool rare_tile;
  {
    u32 s_mx = 0, s_i = 0, s_t = 0, s_x = 0, ev = 0, rare = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const u32 op = w[k], len = op >> 4;
      s_mx += len & bit_mask(MX_BITS, op);
      s_i += len & bit_mask(I_BITS, op);
      s_t += len;
      s_x += len & bit_mask(X_BITS, op);
      ev += bit_test(IEV_BITS, op);
      ev += bit_test(DEV_BITS, op) << 16;
      rare |= bit_mask(RARE_BITS, op);
    }
    rare_tile = __ballot(rare != 0u) != 0ull;
    wave_sums(s_mx, s_i, s_t, s_x, ev, Wmx, Wi, Wt, Wx, Wev);
  }
  u64 Amx = 0, Ai = 0, At = 0, Ax = 0; /* the tile's segments so far */
  u32 Aev = 0;
  while (cur < tile_end) {
    while (re <= cur) { /* skip empty records */
      r++;
      re = op_off[r + 1];
    }
    const bool first = r == r_first;
    const u64 rs = first ? WGA_UNI64(tr.rs) : op_off[r];
    const bool neg = first ? (WGA_UNI32(tr.neg) != 0u) : (strand_neg[r] != 0);
    const u64 seg_end = re < tile_end ? re : tile_end;
    const u32 a = (u32)(cur - tile_start), b = (u32)(seg_end - tile_start);

    /* per-lane partials: 16 ops * (2^28-1) < 2^32, so u32 is exact.  Ops outside the segment are
     * turned into 0M (neutral) on the fly; S / other ops and the first bad op are only worked out when a wave vote
     * says the segment holds any (they end the run with an error anyway). */
    u64 S[5], Sx, St;
    u32 EV, BAD = 0xFFFFFFFFu;
    S[3] = S[4] = 0ull;
    if (!rare_tile && seg_end == tile_end) { /* wave-uniform: what the segments in front leave of the tile */
      S[0] = Wmx - Amx, S[1] = Wi - Ai, St = Wt - At, Sx = Wx - Ax;
      EV = Wev - Aev; /* both counts of the tile are at least those of its first segments: no borrow between the halves */
    } else {
      const u32 span = b - a;
      u32 s_mx = 0, s_i = 0, s_t = 0, s_s = 0, s_o = 0; /* s_t: every op's length — the D bases are what the other classes leave of it */
      u32 s_x = 0;  /* X only: match = s_mx - s_x */
      u32 ev = 0;   /* ins events | del events << 16 */
      u32 bad = 0xFFFFFFFFu;
      u32 rare = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const u32 idx = (u32)(k >> 2) * 256u + (u32)(k & 3) + lane4;
        const u32 op = (idx - a < span) ? w[k] : 0u;
        const u32 len = op >> 4;
        s_mx += len & bit_mask(MX_BITS, op);
        s_i += len & bit_mask(I_BITS, op);
        s_t += len;
        s_x += len & bit_mask(X_BITS, op);
        ev += bit_test(IEV_BITS, op);
        ev += bit_test(DEV_BITS, op) << 16;
        rare |= bit_mask(RARE_BITS, op);
      }
      const bool any_rare = rare_tile && __ballot(rare != 0u) != 0ull;
      if (any_rare) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const u32 idx = (u32)(k >> 2) * 256u + (u32)(k & 3) + lane4;
          u32 op = (idx - a < span) ? w[k] : 0u;
          WGA_PIN(op); /* opaque: no sharing of compare masks with the loop above */
          const u32 cls = op_class(op & 15u), len = op >> 4;
          s_s += cls == CLS_S ? len : 0u;
          s_o += cls == CLS_O ? len : 0u;
          bad = (cls >= CLS_S && idx < bad) ? idx : bad;
        }
      }
      wave_sums(s_mx, s_i, s_t, s_x, ev, S[0], S[1], St, Sx, EV);
      if (any_rare) {
        S[3] = wave_sum_u32_wide(s_s);
        S[4] = wave_sum_u32_wide(s_o);
        BAD = wave_min_u32(bad);
      }
      Amx += S[0], Ai += S[1], At += St, Ax += Sx, Aev += EV;
    }
    S[2] = St - S[0] - S[1] - S[3] - S[4]; /* every op is M-like, I, D, S or other */
    const u64 Smatch = S[0] - Sx;

#if WGA_K1_LANE_STORE
    {
      /* the 11 counters go out from lanes 0..10, one field per lane (v_writelane from the wave-uniform sums):
       * one 88-byte store — or one atomic instruction when the record spans tiles — instead of eleven, and
       * two registers instead of twenty-two */
      const bool whole = rs >= tile_start && re <= tile_end;
      const u64 match = Smatch, mism = S[0] - Smatch;
      const u64 iev = EV & 0xFFFFu, dev = EV >> 16;
      const u64 z = 0;
      u64 v = 0;
      v = lane_put_u64<0u>(v, match, lane);
      v = lane_put_u64<1u>(v, mism, lane);
      v = lane_put_u64<2u>(v, neg ? z : iev, lane);   /* ins_ev.. or inv_ins_ev.. (cigar.rs:667-684) */
      v = lane_put_u64<3u>(v, neg ? z : S[1], lane);
      v = lane_put_u64<4u>(v, neg ? z : dev, lane);
      v = lane_put_u64<5u>(v, neg ? z : S[2], lane);
      v = lane_put_u64<6u>(v, neg ? iev : z, lane);
      v = lane_put_u64<7u>(v, neg ? S[1] : z, lane);
      v = lane_put_u64<8u>(v, neg ? dev : z, lane);
      v = lane_put_u64<9u>(v, neg ? S[2] : z, lane);
      /* inv_event = 1 per '-' record: stored with a whole record, added once by the record's first tile */
      v = lane_put_u64<10u>(v, (neg && (whole || rs >= tile_start)) ? (u64)1 : z, lane);
      u64* const f = (u64*)(counts + r);
      if (lane < 11u) {
        if (whole)
          f[lane] = v; /* the record lives in this tile only: plain stores */
        else if (v)
#ifdef WGA_K1_NO_ATOMICS /* A/B builds only (wrong counts for records across tiles): what the atomics cost */
          f[lane] = v;
#else
          atomicAdd(f + lane, v); /* record spans tiles: counts were zeroed by the launcher */
#endif
      }
      if (lane == 0 && BAD != 0xFFFFFFFFu) atomicMin((u64*)&diag[r].bad_op_idx, tile_start + BAD - rs);
    }
#else
    if (lane == 0) {
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
#endif
#pragma unroll
    for (int c = 0; c < 5; c++) {
      tot[c] += S[c];
      tail[c] = S[c];
    }
    cur = seg_end;
    r++;
    if (cur < tile_end) re = op_off[r + 1];
  }
#if WGA_K1_LANE_STORE
  if (tiles) { /* wga_tile_sum = tot[5], tail[5], rec: one field per lane */
    u64 v = 0;
    v = lane_put_u64<0u>(v, tot[0], lane);
    v = lane_put_u64<1u>(v, tot[1], lane);
    v = lane_put_u64<2u>(v, tot[2], lane);
    v = lane_put_u64<3u>(v, tot[3], lane);
    v = lane_put_u64<4u>(v, tot[4], lane);
    v = lane_put_u64<5u>(v, tail[0], lane);
    v = lane_put_u64<6u>(v, tail[1], lane);
    v = lane_put_u64<7u>(v, tail[2], lane);
    v = lane_put_u64<8u>(v, tail[3], lane);
    v = lane_put_u64<9u>(v, tail[4], lane);
    v = lane_put_u64<10u>(v, (u64)r_first, lane);
    if (lane < 11u) ((u64*)(tiles + g))[lane] = v;
  }
#else
  if (tiles && lane == 0) {
    wga_tile_sum ts;
#pragma unroll
    for (int c = 0; c < 5; c++) {
      ts.tot[c] = tot[c];
      ts.tail[c] = tail[c];
    }
    ts.rec = r_first;
    tiles[g] = ts;
  }
#endif
}

/* 16 ops per lane, each of the 4 loads a fully coalesced 1 KiB.  A 16-byte group is loaded when it starts in front of the
 * stream's end (it may reach up to 12 bytes beyond it, inside the same aligned 16 bytes); what it brings from there becomes
 * 0M, as everything behind the tile's last op */
__device__ __forceinline__ void stat_load_ops(const u32* __restrict__ ops, u64 n_ops, u64 g, u32 lane, u32 (&w)[16]) {
  const u64 tile_start = g * WGA_TILE;
  const u32 nt = tile_start + WGA_TILE < n_ops ? WGA_TILE : (u32)(n_ops - tile_start);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u32 base = ((u32)j * 64u + lane) * 4u;
    u32x4_a16 v = {0u, 0u, 0u, 0u};
    if (base < nt) v = *(const u32x4_a16*)(ops + tile_start + base);
    w[4 * j + 0] = v[0];
    w[4 * j + 1] = v[1];
    w[4 * j + 2] = v[2];
    w[4 * j + 3] = v[3];
  }
  if (nt & 3u) { /* wave-uniform: only the stream's last tile */
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = ((u32)(k >> 2) * 256u + (u32)(k & 3) + lane * 4u < nt) ? w[k] : 0u;
  }
}

/* One wave per tile.  The kernel is bound by its VECTOR instructions — a CU retires one per cycle, and 443 per tile (626 until
 * round 6: profiles/r06_pmc.txt, r05_k1_k5_counters.txt) x 487 536 tiles / 256 CUs is 0.35 of its 0.42 ms — not by its loads (a
 * plain read of the same 2 GB runs at 6.5 TB/s, scripts/micro/read_only.hip; a grid of resident waves that requested the next
 * tile's ops early: 0.586 against 0.555 ms) and not by the atomics of records that span tiles (without them: the same time). */
__global__ __launch_bounds__(256, WGA_K1_BLOCKS) void k_cigar_stat(const u32* __restrict__ ops,
                                                    const u64* __restrict__ op_off,
                                                    const u8* __restrict__ strand_neg, u32 n,
                                                    u64 n_ops, const wga_tile_rec* __restrict__ tile_rec,
                                                    wga_cigar_counts* counts,
                                                    wga_rec_diag* diag, wga_tile_sum* tiles) {
  (void)n;
  const u32 lane = threadIdx.x & 63u;
  const u64 g = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x);
  if (g * WGA_TILE >= n_ops) return; /* wave-uniform; this kernel has no block barrier */
  u32 w[16];
  stat_load_ops(ops, n_ops, g, lane, w);
  cigar_stat_tile(g, w, lane, op_off, strand_neg, n_ops, tile_rec, counts, diag, tiles);
}

/* ============================================================================================ */
/* block-level exclusive scan helpers (256 threads)                                             */
/* ============================================================================================ */
__device__ __forceinline__ u64 block_excl_scan_u64(u64 v, u64* s_w /*[5]*/, u64* total) {
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
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
  u64 q = t + f.t_row(i) + (f.pre_q ? f.pre_q[i] : 0u);
  t_row_off[i] = t;
  q_row_off[i] = q;
}

/* ============================================================================================ */
/* K2: paf2maf gap insertion                                                                    */
/* ============================================================================================ */
#ifndef WGA_DRAIN_MIN
#define WGA_DRAIN_MIN 32u /* queued complex chunks that trigger a drain before the row ends: the rows of K2 take the value from
                             the host (ExpandArgs::drain_min), everything else this one.  64 fills the drain's lanes (fewest
                             instructions) but lets the lines its chunks belong to wait half written in the L2; 16 completes them
                             at once.  Same buffers, one process (scripts/gpu_k2_same_buffers.py): with 2 x 50 MB pools 64 / 32 /
                             16 = 6.19 / 6.44 / 6.66 ms, with 2 x 1 GB pools (the L2 churns with source lines) 8.65 / 8.07 / 7.85 */
#endif
#ifndef WGA_DRAIN_POOL_BYTES
#define WGA_DRAIN_POOL_BYTES (192ull << 20) /* sequence pools beyond this (together) do not stay in the 256 MB Infinity Cache */
#endif
struct RowSrc {
  const u8* fa;  /* sequence pool */
  u64 fa_bytes;  /* pool size (window loads are bounds-checked against it) */
  u64 src_off;   /* start of this record's slice in the pool */
  u64 src_len;   /* slice length as fetched */
  bool rc;       /* read reversed + complemented (utils.rs:83-101) */
  bool safe;     /* every 20-byte window of this row lies inside the pool (rowsrc_prepare) */
  u32 drain_min = WGA_DRAIN_MIN; /* queued complex chunks that trigger a drain before the row ends (see WGA_DRAIN_MIN) */
  const u8* win_base; /* address of slice index sbase (rc: of the mirrored window start) */
};



/* Complement 4 packed bases (utils.rs:86-96) with two 8-entry byte LUTs indexed by the low
 * THREE bits of each base — A=1 C=3 T=4 N=6 G=7 are distinct.  `fold` is the upper-case base a
 * valid byte must equal once its case bit is cleared (0xFF for the unused indices, which no
 * byte & 0xDF can equal), `comp` its complement.  *bad gets a non-zero byte wherever the input
 * is not one of ACGTNacgtn. */
__device__ __forceinline__ u32 comp4(u32 x, u32* bad) {
  const u32 sel = x & 0x07070707u;
  /* index:      7     6     5     4        3     2     1     0   */
  const u32 fold = byte_perm(0x474EFF54u, 0x43FF41FFu, sel); /* G N - T | C - A - */
  const u32 comp = byte_perm(0x434EFF41u, 0x47FF54FFu, sel); /* C N - A | G - T - */
  *bad = (x & 0xDFDFDFDFu) ^ fold;
  return comp | (x & 0x20202020u);
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

/* A 16-byte source window: output byte j of a chunk <-> slice index S + j (S may be negative /
 * beyond the slice for bytes outside the piece — those are masked by the caller).  Split in two
 * so that several windows can be in flight: win_issue only issues the load (ONE byte-aligned
 * global_load_dwordx4), win_finish — for rc — reverses + complements and flags invalid bases
 * (non-zero byte).  `off` = S - sbase is a small i32; the 64-bit part of the address is
 * wave-uniform (RowSrc::win_base).  Rows whose slice sits >= 32 bytes inside the pool (`safe`,
 * uniform) load unchecked; rows at a pool edge take guarded byte loads. */
struct WinRaw {
  u32 v[4];
};

__device__ __forceinline__ void win_issue(const RowSrc& src, u64 sbase, int off, int pa, int pb,
                                          WinRaw& r) {
  if (src.safe) {
    /* pointer arithmetic only (no integer round trip): keeps this a global_load, not flat */
    const int sgn = src.rc ? -1 : 0; /* uniform */
    const u8* p = src.win_base + (i64)((off ^ sgn) - sgn);
    u32x4_a1 v = *(const u32x4_a1*)p;
    r.v[0] = v[0];
    r.v[1] = v[1];
    r.v[2] = v[2];
    r.v[3] = v[3];
  } else { /* pool edge: guarded byte loads, only for the bytes of the piece */
    const i64 S = (i64)sbase + off;
    const i64 P = src.rc ? (i64)src.src_off + (i64)src.src_len - 16 - S : (i64)src.src_off + S;
    u64 lo = 0, hi = 0; /* no dynamically indexed array: that would force WinRaw into scratch */
    for (int j = pa; j < pb; j++) {
      int vj = src.rc ? 15 - j : j; /* position inside the address-ordered window */
      i64 idx = P + vj;
      u64 byte = (idx >= 0 && (u64)idx < src.fa_bytes) ? (u64)src.fa[idx] : 0ull;
      if (vj < 8)
        lo |= byte << (8 * vj);
      else
        hi |= byte << (8 * (vj - 8));
    }
    r.v[0] = (u32)lo;
    r.v[1] = (u32)(lo >> 32);
    r.v[2] = (u32)hi;
    r.v[3] = (u32)(hi >> 32);
  }
}

__device__ __forceinline__ void win_finish(const RowSrc& src, const WinRaw& r, u32 W[4],
                                           u32 inv[4]) {
  if (src.rc) {
    W[0] = comp4(bswap32(r.v[3]), &inv[0]);
    W[1] = comp4(bswap32(r.v[2]), &inv[1]);
    W[2] = comp4(bswap32(r.v[1]), &inv[2]);
    W[3] = comp4(bswap32(r.v[0]), &inv[3]);
  } else {
    W[0] = r.v[0];
    W[1] = r.v[1];
    W[2] = r.v[2];
    W[3] = r.v[3];
    inv[0] = inv[1] = inv[2] = inv[3] = 0u;
  }
}

/* bytes [0, n) of a 16-byte vector set: table of 17 masks, built once per block in LDS */
__device__ __forceinline__ void build_lowmask(u32x4_a16* lm) {
  if (threadIdx.x < 17u) {
    u32x4_a16 m;
    for (int d = 0; d < 4; d++) m[d] = bytemask(0, (int)threadIdx.x - 4 * d);
    lm[threadIdx.x] = m;
  }
}

/* o = bytes [pa, pb) from W, the rest unchanged */
__device__ __forceinline__ void merge16(u32 o[4], const u32 W[4], int pa, int pb,
                                        const u32x4_a16* lm) {
  const u32x4_a16 hi = lm[pb], lo = lm[pa];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    u32 m = hi[d] & ~lo[d];
    o[d] = (o[d] & ~m) | (W[d] & m);
  }
}
__device__ __forceinline__ void merge_dash(u32 o[4], int pa, int pb, const u32x4_a16* lm) {
  const u32 dash[4] = {0x2D2D2D2Du, 0x2D2D2D2Du, 0x2D2D2D2Du, 0x2D2D2D2Du};
  merge16(o, dash, pa, pb, lm);
}

/* InvalidBase: first offender in reversed order = smallest q' index */
__device__ __forceinline__ void flag_bad_bases(const u32 inv[4], int pa, int pb, i64 S,
                                               const u32x4_a16* lm, u64* bad_base_pos) {
  const u32x4_a16 hi = lm[pb], lo = lm[pa];
  if (((inv[0] & hi[0] & ~lo[0]) | (inv[1] & hi[1] & ~lo[1]) | (inv[2] & hi[2] & ~lo[2]) |
       (inv[3] & hi[3] & ~lo[3])) == 0u)
    return;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    u32 bad = inv[d] & hi[d] & ~lo[d];
    if (bad) {
      int j = 4 * d + ((__ffsll((unsigned long long)bad) - 1) >> 3);
      atomicMin(bad_base_pos, (u64)(S + j));
    }
  }
}

/* Row description shared by the emitters.  The row's events inside [c0, c0+N) are entries
 * [ga, gb) of a compacted list (G_col = tile-relative start column, G_cum = exclusive prefix of
 * gap bases — an entry's gap length is G_cum[i+1]-G_cum[i], entry gb is readable — and G_adj =
 * exclusive prefix of the source adjustment: gap bases minus skipped source bases, as wrapping
 * u32; for paf2maf rows G_adj == G_cum).  A non-gap column c reads slice index
 *     sbase + (c - c_org) - (adj before c - gcum_a)          (gcum_a = adj at c_org) */
struct RowDesc {
  u32 c_org;
  const u32* G_col;
  const u32* G_cum;
  const u32* G_adj;
  int ga, gb;
  u32 gcum_a;
  u64 sbase;
  const u32x4_a16* lowmask;
  const u32* tbl; /* granule table: T[j] = events (tile-wide index) that start before granule j */
  u32 tsh;        /* bit offset of this row's counts inside a table word */
  u32 gsh;        /* log2 of the granule width in columns (>= 4) */
  u32* queue;     /* WGA_QCAP words per wave */
};

/* generic piece walk of one chunk from an arbitrary state (any number of pieces) */
__device__ __forceinline__ void emit_walk(u32 o[4], u32 c, u32 c_end, u32 cz, int i, bool in_gap,
                                          u32 gap_end, u32 cum, const RowDesc& rd,
                                          const RowSrc& src, u64* bad_base_pos) {
  while (c < c_end) {
    if (in_gap) {
      u32 pe = gap_end < c_end ? gap_end : c_end;
      merge_dash(o, (int)(c - cz), (int)(pe - cz), rd.lowmask);
      c = pe;
      in_gap = false;
    } else {
      u32 next_gs = (i + 1 < rd.gb) ? rd.G_col[i + 1] : 0xFFFFFFFFu;
      u32 pe = next_gs < c_end ? next_gs : c_end;
      if (pe > c) {
        const int pa = (int)(c - cz), pb = (int)(pe - cz);
        const int off = (int)(cz - rd.c_org) - (int)(cum - rd.gcum_a);
        WinRaw raw;
        u32 W[4], inv[4];
        win_issue(src, rd.sbase, off, pa, pb, raw);
        win_finish(src, raw, W, inv);
        if (src.rc) flag_bad_bases(inv, pa, pb, (i64)rd.sbase + off, rd.lowmask, bad_base_pos);
        merge16(o, W, pa, pb, rd.lowmask);
        c = pe;
      }
      if (c < c_end) { /* c == start of entry i+1 */
        i++;
        u32 gs = rd.G_col[i];
        u32 gl = rd.G_cum[i + 1] - rd.G_cum[i];
        if (gl) {
          in_gap = true;
          gap_end = gs + gl;
        }
        cum = rd.G_adj[i + 1];
      }
    }
  }
}


#define WGA_TBL_SHIFT 4u                          /* granule = 16 columns */
#ifndef WGA_TBL_COLS
#define WGA_TBL_COLS 16384u /* widest tile the 16-column granule table covers (wider tiles use 32-, 64-... column granules);
                               32768 costs 4 KB more LDS and with it the sixth block per CU, and measures no faster */
#endif
#define WGA_TBL_N (WGA_TBL_COLS >> WGA_TBL_SHIFT) /* 1024 granules (+2 sentinels) */
#ifndef WGA_EMIT_U
#define WGA_EMIT_U 4 /* chunks in flight per lane */
#endif
#ifndef WGA_SOLO_BYTES
#define WGA_SOLO_BYTES 65536u /* rows up to this many bytes are emitted wave by wave, longer ones by the block */
#endif
#ifndef WGA_SPLIT_BYTES
#define WGA_SPLIT_BYTES 8192u /* ... in two halves beyond this */
#endif
#ifndef WGA_OWNER_SETUP
#define WGA_OWNER_SETUP 1 /* 1: only the wave that owns a row piece reads the row's source / destination fields */
#endif
#define WGA_QCAP (64u * (WGA_EMIT_U + 1u)) /* per-wave queue of complex chunks: < 64 left over + one iteration's pushes */

/* Granule table: one 16-bit field per row (target row = low half, query row = high half of a
 * word; `tsh` selects).  After the exclusive scan entry j holds, for the events of that row,
 *   bits 0-10  how many start in granules < j (tile-wide entry index of the first one at / after j)
 *   bit  11    COVER: granule j-1 begins inside a gap that started in an earlier granule
 *   bit  12    FULL:  ... and that gap runs through the end of granule j-1
 * (COVER / FULL of granule j are read from entry j+1.)  They come from +1 / -1 marks at the
 * first covered granule and after the last one; a field's prefix sums are never negative, so the
 * borrows that packed two's-complement adds take from the neighbouring fields cancel out. */
#define WGA_TBL_CNT 0x7FFu
#define WGA_TBL_COVER 0x800u
#define WGA_TBL_FULL 0x1000u
__device__ __forceinline__ void tbl_mark_event(u32* tbl, u32 gs, u32 gl, u32 gsh, u32 tsh) {
  const u32 js1 = (gs >> gsh) + 1u;
  atomicAdd(&tbl[js1 - 1u], 1u << tsh);
  const u32 ge = gs + gl;
  const u32 jc = (ge + (1u << gsh) - 1u) >> gsh, jf = ge >> gsh; /* jf <= jc */
  if (jc > js1) {
    const bool any_full = jf > js1;
    atomicAdd(&tbl[js1], (any_full ? (WGA_TBL_COVER | WGA_TBL_FULL) : WGA_TBL_COVER) << tsh);
    if (any_full && jf != jc) atomicAdd(&tbl[jf], (0u - WGA_TBL_FULL) << tsh);
    atomicAdd(&tbl[jc], (0u - ((any_full && jf == jc) ? (WGA_TBL_COVER | WGA_TBL_FULL) : WGA_TBL_COVER)) << tsh);
  }
}
/* exclusive scan of the raw per-granule marks, in place; entries [WGA_TBL_N], [WGA_TBL_N+1] = total */
__device__ __forceinline__ void tbl_scan(u32* tbl, u32* s_w4) {
  const u32 tid = threadIdx.x;
  u32 v[WGA_TBL_N / WGA_BLOCK], sum = 0;
#pragma unroll
  for (u32 e = 0; e < WGA_TBL_N / WGA_BLOCK; e++) {
    v[e] = tbl[tid * (WGA_TBL_N / WGA_BLOCK) + e];
    sum += v[e];
  }
  const u32 lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  const u32 inc = wave_incl_scan_u32(sum);
  if (lane == 63u) s_w4[wave] = inc; /* callers have a barrier between the last use of s_w4 and this */
  __syncthreads();
  u32 run = inc - sum, tot = 0;
#pragma unroll
  for (u32 w = 0; w < 4; w++) {
    const u32 x = s_w4[w];
    run += w < wave ? x : 0u;
    tot += x;
  }
#pragma unroll
  for (u32 e = 0; e < WGA_TBL_N / WGA_BLOCK; e++) {
    tbl[tid * (WGA_TBL_N / WGA_BLOCK) + e] = run;
    run += v[e];
  }
  if (tid == WGA_BLOCK - 1) tbl[WGA_TBL_N] = tbl[WGA_TBL_N + 1] = tot;
}

/* Chunks are the 16-column granules of the tile-relative column space (so the granule table
 * classifies them exactly); their output address is whatever it is — stores are byte-aligned
 * 16-byte stores.  Only the first / last granule of a row span can be partial. */
struct ChunkGeom {
  u8* p;      /* address of byte 0 of the granule (may lie before the row for a head granule) */
  u32 a0, b0; /* valid bytes [a0, b0) of the 16 */
  u32 cz;     /* column of byte 0 */
  u32 c, c_end;
};
struct RowGeom {
  u8* base;    /* address of column 16*j0 */
  u32 j0;      /* first granule */
  u32 head;    /* c0 & 15 */
  u32 nchunks; /* granules touched by [c0, c0+N) */
  u32 last_b0; /* valid end of the last granule (1..16) */
};
__device__ __forceinline__ RowGeom row_geom(u8* dst, u32 N, u32 c0) {
  RowGeom r;
  r.head = c0 & 15u;
  r.j0 = c0 >> 4;
  r.base = dst - r.head; /* pointer arithmetic: stores stay global_store */
  r.nchunks = (r.head + N + 15u) >> 4;
  r.last_b0 = ((r.head + N - 1u) & 15u) + 1u;
  return r;
}
__device__ __forceinline__ ChunkGeom chunk_geom(const RowGeom& r, u32 rel) {
  ChunkGeom g;
  g.p = r.base + (rel << 4);
  g.a0 = rel == 0u ? r.head : 0u;
  g.b0 = rel == r.nchunks - 1u ? r.last_b0 : 16u;
  g.cz = (r.j0 + rel) << 4;
  g.c = g.cz + g.a0;
  g.c_end = g.cz + g.b0;
  return g;
}
__device__ __forceinline__ void chunk_store(const ChunkGeom& g, const u32 o[4]) {
  if (g.a0 == 0u && g.b0 == 16u) {
    u32x4_a1 v = {o[0], o[1], o[2], o[3]};
    *(u32x4_a1*)g.p = v;
  } else { /* partial chunk at a row / tile edge: byte stores, never read-modify-write */
    u8* p = g.p;
    for (u32 j = g.a0; j < g.b0; j++) {
      u32 d = j >> 2;
      u32 word = d == 0 ? o[0] : d == 1 ? o[1] : d == 2 ? o[2] : o[3];
      p[j] = (u8)(word >> (8u * (j & 3u)));
    }
  }
}

/* i = last entry in [ga, gb) whose column is <= c, or ga-1: two table reads and a walk over the
 * entries of one granule (0..1 entries when the granule is 16 columns wide). */
__device__ __forceinline__ int find_entry(const RowDesc& rd, u32 c) {
  const int ga = rd.ga, gb = rd.gb;
  const u32 j = c >> rd.gsh;
  int k = (int)((rd.tbl[j] >> rd.tsh) & WGA_TBL_CNT);
  const int hi = (int)((rd.tbl[j + 1] >> rd.tsh) & WGA_TBL_CNT);
  while (k < hi && rd.G_col[k] <= c) k++;
  k -= 1;
  return k < ga ? ga - 1 : (k >= gb ? gb - 1 : k);
}

/* A chunk that touches an event boundary, straight-line for rows whose windows need no bounds
 * checks:  [gap0 rest] copy0 | gap1 | copy1  with two source windows, assembled with byte masks
 * from the low-mask table; a third event inside the 16 columns continues in emit_walk.  Every
 * lane of a drain runs the same instructions whatever its chunk looks like (pieces may be
 * empty): the drains mix all shapes, so branches would only add their overhead. */
__device__ __forceinline__ u32 bfi32(u32 mask, u32 a, u32 b) { return (a & mask) | (b & ~mask); }

/* the two buffers of a row (see emit_row): source windows, biased so that small negative and
 * downward (rc) offsets stay positive, and the row's output granules */
struct RowBufs {
  BufRsrc lbuf, sbuf;
  u32 sgn, kbias;
};
__device__ __forceinline__ u32 rowbuf_loff(const RowBufs& b, int off) {
  return (((u32)off ^ b.sgn) - b.sgn) + b.kbias;
}

__device__ __forceinline__ void complex_chunk(const ChunkGeom& g, u32 rel, const RowDesc& rd,
                                              const RowSrc& src, const RowBufs& rb,
                                              u64* bad_base_pos) {
  const int ga = rd.ga, gb = rd.gb;
  const u32 c = g.c, c_end = g.c_end, cz = g.cz;
  const int i = find_entry(rd, c);
  const bool has0 = i >= ga;
  const int ic = has0 ? i : ga; /* always a readable index */
  const u32 gs0 = rd.G_col[ic], cum0a = rd.G_cum[ic], cum0b = rd.G_cum[ic + 1];
  const u32 adj0b = rd.G_adj[ic + 1];
  const u32 gl0 = cum0b - cum0a;
  const bool in_gap0 = has0 && (c - gs0 < gl0);
  const u32 adj0 = has0 ? adj0b : rd.gcum_a;
  const int n1 = i + 1;
  const u32 gs1 = n1 < gb ? rd.G_col[n1] : 0xFFFFFFFFu;
  const u32 gl1 = rd.G_cum[n1 + 1] - rd.G_cum[n1]; /* two sentinels: readable up to gb + 1 */
  const u32 adj1 = rd.G_adj[n1 + 1];
  const u32 gs2 = n1 + 1 < gb ? rd.G_col[n1 + 1] : 0xFFFFFFFFu;
  const bool hasB = gs1 < c_end;
  const u32 g0e = gs0 + gl0;
  const u32 a1 = in_gap0 ? (g0e < c_end ? g0e : c_end) : c;       /* copy piece 0 = [a1, b1) */
  const u32 b1 = hasB ? gs1 : c_end;
  const u32 g1e = gs1 + gl1;
  const u32 e1 = hasB ? (g1e < c_end ? g1e : c_end) : c_end;      /* gap 1 = [b1, e1)        */
  const u32 b2 = hasB ? (gs2 < c_end ? gs2 : c_end) : c_end;      /* copy piece 1 = [e1, b2) */
  const int offz = (int)(cz - rd.c_org);
  const int off0 = offz - (int)(adj0 - rd.gcum_a), off1 = offz - (int)(adj1 - rd.gcum_a);
  u32 o[4];
  if (src.safe) {
    WinRaw r0, r1;
    buf_load16(rb.lbuf, b1 > a1 ? rowbuf_loff(rb, off0) : WGA_BUF_OOB, r0.v);
    buf_load16(rb.lbuf, b2 > e1 ? rowbuf_loff(rb, off1) : WGA_BUF_OOB, r1.v);
    const u32x4_a16 La1 = rd.lowmask[a1 - cz], Lb1 = rd.lowmask[b1 - cz], Le1 = rd.lowmask[e1 - cz],
                    Lb2 = rd.lowmask[b2 - cz];
    u32 W0[4], W1[4], inv0[4], inv1[4];
    win_finish(src, r0, W0, inv0);
    win_finish(src, r1, W1, inv1);
    u32 bad = 0u;
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const u32 m0 = Lb1[d] & ~La1[d], m1 = Lb2[d] & ~Le1[d];
      const u32 md = bfi32(Lb1[d], La1[d], Le1[d]); /* [0, a1) + [b1, e1): bytes below c are never stored */
      o[d] = bfi32(m0, W0[d], bfi32(m1, W1[d], md & 0x2D2D2D2Du));
      bad |= (inv0[d] & m0) | (inv1[d] & m1);
    }
    if (bad) { /* rare: report the first invalid base (utils.rs:97) */
      flag_bad_bases(inv0, (int)(a1 - cz), (int)(b1 - cz), (i64)rd.sbase + off0, rd.lowmask, bad_base_pos);
      flag_bad_bases(inv1, (int)(e1 - cz), (int)(b2 - cz), (i64)rd.sbase + off1, rd.lowmask, bad_base_pos);
    }
    if (b2 < c_end) /* a third event inside 16 columns: rare, generic walk from there */
      emit_walk(o, b2, c_end, cz, n1, false, 0u, adj1, rd, src, bad_base_pos);
  } else { /* a row at a pool edge: guarded byte loads, generic walk */
    o[0] = o[1] = o[2] = o[3] = 0u;
    emit_walk(o, c, c_end, cz, i, in_gap0, g0e, adj0, rd, src, bad_base_pos);
  }
  const bool whole = g.a0 == 0u && g.b0 == 16u;
  buf_store16(rb.sbuf, whole ? rel << 4 : WGA_BUF_OOB, o);
  if (!whole) chunk_store(g, o); /* a row / tile edge: byte stores */
}


/* RC = the row is read reverse-complemented: a compile-time copy of src.rc, so that each of the
 * two instantiations carries only its own window post-processing */
template <bool RC>
__device__ __forceinline__ void emit_row_t(u8* dst, u32 N, u32 c0, const RowDesc& rd,
                                           const RowSrc& src_in, u32 tid, u32 nthreads,
                                           u64* bad_base_pos) {
  if (N == 0) return;
  RowSrc src = src_in;
  src.rc = RC;
  const u32 lane = tid & 63u;
  const RowGeom rg = row_geom(dst, N, c0);
  u32* const queue = rd.queue + WGA_WAVE_ID(threadIdx.x) * WGA_QCAP; /* the wave's own, whoever it works with */
  u32 qn = 0; /* wave-uniform queue length */
  const u32 per_it = nthreads * WGA_EMIT_U;
  const u32 niter = (rg.nchunks + per_it - 1) / per_it;
  /* chunks [lo_full, lo_full + n_full) are whole 16-column granules */
  const u32 lo_full = rg.head == 0u ? 0u : 1u;
  const u32 n_full = rg.nchunks - lo_full - (rg.last_b0 == 16u ? 0u : 1u); /* may wrap to "none" */
  const bool any_full = rg.nchunks >= lo_full + (rg.last_b0 == 16u ? 0u : 1u) + 1u;
  const int koff = (int)(rd.gcum_a - rd.c_org); /* window offset of a chunk = cz + koff - adj */
  const bool row_fast = src.safe;
  const bool fast_ok = row_fast && any_full;
  /* buffers of the fast path: the source windows of this row relative to its first window (for
   * rc the windows walk down from it: offsets are biased by 2^31), and the row's output granules */
  RowBufs rb;
  rb.sgn = src.rc ? 0xFFFFFFFFu : 0u;
  rb.kbias = src.rc ? 0x80000000u : 64u;
  rb.lbuf = buf_make(src.win_base - (i64)rb.kbias, 0xFFFFFFF0u);
  rb.sbuf = buf_make(rg.base, rg.nchunks << 4);
#pragma nounroll
  for (u32 it = 0; it < niter; it++) {
    /* Fast path: a whole granule that no event of this row touches — plain copy — or that lies
     * inside one gap — dashes — read off the granule table (two words) plus one adjustment.
     * WGA_EMIT_U chunks per lane go through it together (lookups, then loads, then stores), all
     * of it branch-free: lanes without a candidate use an out-of-range buffer offset.
     * Everything else is deferred to the queue. */
    u32 rel[WGA_EMIT_U], loff[WGA_EMIT_U];
    bool act[WGA_EMIT_U], cand[WGA_EMIT_U], dash[WGA_EMIT_U];
    u32 raw[WGA_EMIT_U][4];
#pragma unroll
    for (int u = 0; u < WGA_EMIT_U; u++) {
      rel[u] = (it * WGA_EMIT_U + (u32)u) * nthreads + tid;
      act[u] = rel[u] < rg.nchunks;
      const u32 relc = act[u] ? rel[u] : 0u;
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
      cand[u] = (bool)((int)fast_ok & (int)(rel[u] - lo_full < n_full) & (int)(((w0 ^ w1) & WGA_TBL_CNT) == 0u) &
                       (int)(st != WGA_TBL_COVER));
      const u32 off = cz + (u32)koff - adj; /* slice index of the granule relative to sbase, >= 0 */
      loff[u] = ((int)cand[u] & (int)!dash[u]) ? rowbuf_loff(rb, (int)off) : WGA_BUF_OOB;
    }
#pragma unroll
    for (int u = 0; u < WGA_EMIT_U; u++) buf_load16(rb.lbuf, loff[u], raw[u]);
    bool cxs[WGA_EMIT_U];
#pragma unroll
    for (int u = 0; u < WGA_EMIT_U; u++) {
      u32 o[4], inv[4];
      WinRaw wr;
      wr.v[0] = raw[u][0];
      wr.v[1] = raw[u][1];
      wr.v[2] = raw[u][2];
      wr.v[3] = raw[u][3];
      win_finish(src, wr, o, inv);
      const bool bad = ((inv[0] | inv[1] | inv[2] | inv[3]) != 0u) && !dash[u];
#pragma unroll
      for (int d = 0; d < 4; d++) o[d] = dash[u] ? 0x2D2D2D2Du : o[d];
      const bool st_ok = cand[u] && !bad; /* an invalid base: the complex path finds and reports it */
        buf_store16(rb.sbuf, st_ok ? rel[u] << 4 : WGA_BUF_OOB, o);
      cxs[u] = act[u] && !st_ok;
    }
    /* compact the complex chunks into the wave queue */
#pragma unroll
    for (int u = 0; u < WGA_EMIT_U; u++) {
      const u64 m = __ballot(cxs[u]);
      if (m) {
        if (cxs[u]) queue[qn + lane_rank(m, lane)] = rel[u];
        qn += (u32)__popcll(m);
      }
    }
    WGA_WAVE_SYNC();
    /* drain the queue 64 chunks at a time, and whatever is left when the row ends */
    const bool last = it + 1 >= niter;
    while (qn >= src.drain_min || (last && qn > 0u)) {
      const u32 take = qn < 64u ? qn : 64u;
      qn -= take;
      if (lane < take) {
        const u32 qrel = queue[qn + lane];
        complex_chunk(chunk_geom(rg, qrel), qrel, rd, src, rb, bad_base_pos);
      }
      WGA_WAVE_SYNC(); /* the drained slots are rewritten by the next pushes */
    }
  }
}

__device__ __forceinline__ void emit_row(u8* dst, u32 N, u32 c0, const RowDesc& rd,
                                         const RowSrc& src, u32 tid, u32 nthreads,
                                         u64* bad_base_pos) {
  if (src.rc)
    emit_row_t<true>(dst, N, c0, rd, src, tid, nthreads, bad_base_pos);
  else
    emit_row_t<false>(dst, N, c0, rd, src, tid, nthreads, bad_base_pos);
}

/* finish a RowSrc: the wave-uniform 64-bit part of every window address of this row */
__device__ __forceinline__ void rowsrc_prepare(RowSrc& s, u64 sbase) {
  s.safe = s.src_off >= 16 && s.src_off + s.src_len + 16 <= s.fa_bytes;
  s.win_base = s.rc ? s.fa + s.src_off + s.src_len - 16 - sbase : s.fa + s.src_off + sbase;
}

/* tail of a row when the fetched slice is longer than the CIGAR consumes: plain copy.  `zero2`
 * = two zero words in LDS (an empty granule table), `dummy` = any readable LDS array. */
__device__ __forceinline__ void emit_tail(u8* dst, u64 n, u64 sbase, RowSrc src,
                                          const u32x4_a16* lowmask, u32* queue, const u32* zero2,
                                          const u32* dummy, u32 tid, u32 nthreads,
                                          u64* bad_base_pos) {
  u64 done = 0;
  while (done < n) {
    u64 m = n - done;
    if (m > (1ull << 30)) m = 1ull << 30;
    RowDesc rd;
    rd.c_org = 0u;
    rd.G_col = rd.G_cum = dummy;
    rd.G_adj = zero2; /* the fast path reads G_adj[0] */
    rd.ga = rd.gb = 0;
    rd.gcum_a = 0u;
    rd.sbase = sbase + done;
    rd.lowmask = lowmask;
    rd.tbl = zero2;
    rd.tsh = 0u;
    rd.gsh = 31u;
    rd.queue = queue;
    rowsrc_prepare(src, rd.sbase);
    emit_row(dst + done, (u32)m, 0u, rd, src, tid, nthreads, bad_base_pos);
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

/* Per-record geometry gathered once (k_rec_desc) so that the expand kernel fetches one compact
 * struct per segment instead of ten scattered values; per-tile base sums of the record that
 * continues into a tile (k_tile_base) so that it never walks tile summaries itself. */
struct wga_rec_desc {
  u64 t_row_off, q_row_off;
  u64 t_src_off, t_src_len, q_src_off, q_src_len;
  u64 I_total, D_total, L;
  u64 neg;
};
struct wga_tile_base {
  u64 mx, i, d;
};
/* Everything the expand kernel needs before it can start on a tile, in one 128-byte record that
 * every wave fetches with a single 32-lane load at the top of the kernel (no dependent round
 * trips: tile summary -> record index -> offsets / descriptor).  Dword layout is fixed: the
 * kernel picks fields out of lanes with v_readlane. */
struct wga_tile_desc {
  u64 tile_cols;        /* dwords 0-1   columns of the tile (M = X I D bases)                  */
  u32 rec, neg;         /* 2, 3         record of the tile's first op; its strand              */
  u64 b_mx, b_i, b_d;   /* 4-9          class sums of that record before the tile              */
  u64 rs, re;           /* 10-13        op_off[rec], op_off[rec + 1]                           */
  u64 t_row_off, q_row_off, t_src_off, t_src_len, q_src_off, q_src_len, I_total, D_total, L; /* 14-31 */
};

__global__ __launch_bounds__(256) void k_rec_desc(u32 n, const wga_cigar_counts* counts,
                                                  const u8* strand_neg, const u64* t_src_off,
                                                  const u64* t_src_len, const u64* q_src_off,
                                                  const u64* q_src_len, const u64* t_row_off,
                                                  const u64* q_row_off, wga_rec_desc* out) {
  u32 r = blockIdx.x * 256u + threadIdx.x;
  if (r >= n) return;
  const wga_cigar_counts cn = counts[r];
  wga_rec_desc d;
  d.t_row_off = t_row_off[r];
  d.q_row_off = q_row_off[r];
  d.t_src_off = t_src_off[r];
  d.t_src_len = t_src_len[r];
  d.q_src_off = q_src_off[r];
  d.q_src_len = q_src_len[r];
  d.I_total = cn.ins_bp + cn.inv_ins_bp;
  d.D_total = cn.del_bp + cn.inv_del_bp;
  d.L = cn.match + cn.mismatch + d.I_total + d.D_total;
  /* bit 0 strand; bits 1 / 2: the target / query slice is longer than the CIGAR consumes (a tail to append) — lets the
   * row kernels skip the two tail jobs of a record, which are empty in any consistent PAF, without reading a field */
  /* bit 3: a slice is shorter than its CIGAR consumes — the only case in which String::insert_str can be asked to insert beyond
   * the end (cigar.rs:507,513); the window kernel checks the gap lists of such records only */
  d.neg = (strand_neg[r] != 0 ? 1u : 0u) | (d.t_src_len + d.I_total > d.L ? 2u : 0u) | (d.q_src_len + d.D_total > d.L ? 4u : 0u) |
          ((d.t_src_len + d.I_total < d.L || d.q_src_len + d.D_total < d.L) ? 8u : 0u);
  out[r] = d;
}

/* One thread per tile: class sums of the tile's first record before the tile = tail of the tile
 * where the record starts + totals of the tiles in between, plus everything else the expand
 * kernel wants to know up front (wga_tile_desc).  Walk-backs over more than 64 tiles (records
 * beyond 64 kop) are summed by the whole wave, one such tile at a time. */
/* pseudo != 0 (pafpseudo's rows, wga_kernels_k2s.h): S ops count as I (both skip query bases without a column), and a tile's
 * "columns" are everything it makes a row kernel walk (M = X I D S). */
__global__ __launch_bounds__(256) void k_tile_base(const u64* op_off, u64 n_ops,
                                                   const wga_tile_sum* tiles,
                                                   const wga_rec_desc* recs, wga_tile_desc* descs, int pseudo) {
  const u32 lane = threadIdx.x & 63u;
  const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 tile_start = g * WGA_TILE;
  const bool valid = tile_start < n_ops;
  wga_tile_sum ts;
  u64 rs = 0, re = 0, g0 = 0;
  if (valid) {
    ts = tiles[g];
    rs = op_off[ts.rec];
    re = op_off[ts.rec + 1];
  }
  u64 p_mx = 0, p_i = 0, p_d = 0;
  bool far = false;
  if (valid && rs < tile_start) {
    g0 = rs / WGA_TILE;
    if (g - g0 <= 64) {
      for (u64 k = g0; k < g; k++) {
        const u64* v = (k == g0) ? tiles[k].tail : tiles[k].tot;
        p_mx += v[CLS_MX];
        p_i += v[CLS_I] + (pseudo ? v[CLS_S] : 0ull);
        p_d += v[CLS_D];
      }
    } else {
      far = true;
    }
  }
  u64 m = __ballot(far);
  while (m) {
    const int src = (int)__builtin_ctzll(m);
    const u64 gg = __shfl(g, src), gg0 = __shfl(g0, src);
    u64 a_mx = 0, a_i = 0, a_d = 0;
    for (u64 k = gg0 + lane; k < gg; k += 64) {
      const u64* v = (k == gg0) ? tiles[k].tail : tiles[k].tot;
      a_mx += v[CLS_MX];
      a_i += v[CLS_I] + (pseudo ? v[CLS_S] : 0ull);
      a_d += v[CLS_D];
    }
    a_mx = wave_sum_u64(a_mx);
    a_i = wave_sum_u64(a_i);
    a_d = wave_sum_u64(a_d);
    if ((int)lane == src) {
      p_mx = a_mx;
      p_i = a_i;
      p_d = a_d;
    }
    m &= m - 1;
  }
  if (!valid) return;
  const wga_rec_desc rd = recs[ts.rec];
  wga_tile_desc d;
  d.tile_cols = ts.tot[CLS_MX] + ts.tot[CLS_I] + ts.tot[CLS_D] + (pseudo ? ts.tot[CLS_S] : 0ull);
  d.rec = (u32)ts.rec;
  d.neg = (u32)rd.neg;
  d.b_mx = p_mx;
  d.b_i = p_i;
  d.b_d = p_d;
  d.rs = rs;
  d.re = re;
  d.t_row_off = rd.t_row_off;
  d.q_row_off = rd.q_row_off;
  d.t_src_off = rd.t_src_off;
  d.t_src_len = rd.t_src_len;
  d.q_src_off = rd.q_src_off;
  d.q_src_len = rd.q_src_len;
  d.I_total = rd.I_total;
  d.D_total = rd.D_total;
  d.L = rd.L;
  descs[g] = d;
}

struct ExpandArgs {
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
  int force_slow;
  int no_table; /* test knob: 256-column granules (the coarse-table path of very wide tiles) */
  u32 drain_min; /* RowSrc::drain_min of the rows */
  const u32* tile_count; /* k_paf2maf_expand_list: the blocks loop over tile_list[0 .. *tile_count) */
  const u32* tile_list;
  u32 n_rec;    /* records of the batch (op_off has n_rec + 1 entries) */
  u32 job_tiles;   /* streaming kernel: consecutive tiles one wave walks as one stream */
};

#ifndef WGA_K2_BLOCKS
#define WGA_K2_BLOCKS 6 /* blocks per CU the register budget of k_paf2maf_expand is sized for: 80 VGPRs, no scratch,
                           26 KB of LDS.  Five (93 VGPRs, 30 KB with the 32768-column table): 7.24-7.6 ms, six: 6.88 ms */
#endif
__device__ __forceinline__ void expand_tile_v1(const ExpandArgs& a, const u64 g) {
  __shared__ u32 s_bnd[3][2];            /* (column, I | D << 16 gap-op counts) before the op where a record
                                            ends inside the tile: [0] the first such op (written in phase A),
                                            [1], [2] later ones (rebuilt on demand)               */
  __shared__ u32 s_tot[2];               /* ... and at the end of the tile                    */
  __shared__ u32 s_tg_col[WGA_TILE + 2]; /* target-row gaps (I ops): start column             */
  __shared__ u32 s_tg_cum[WGA_TILE + 2]; /*                           gap bases before        */
  __shared__ u32 s_qg_col[WGA_TILE + 2]; /* query-row gaps (D ops)                            */
  __shared__ u32 s_qg_cum[WGA_TILE + 2];
  __shared__ u32 s_zero2[2];
  __shared__ u32 s_w4[16];
  __shared__ u32x4_a16 s_lowmask[17];
  __shared__ u32 s_tbl[WGA_TBL_N + 2];   /* entries before each 16-column granule: I | D<<16  */
  __shared__ u32 s_queue[4 * WGA_QCAP];  /* per-wave queues of complex chunks                 */

  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  const u64 tile_start = g * WGA_TILE;
  const u64 tile_end = tile_start + WGA_TILE < a.n_ops ? tile_start + WGA_TILE : a.n_ops;
  const u32 nt = (u32)(tile_end - tile_start);
  build_lowmask(s_lowmask);
  /* lane k of every wave holds dword k of this tile's wga_tile_desc: one VGPR, no dependent
   * loads; fields are picked out with v_readlane when they are needed */
  u32 pre = 0u;
  if (lane < 32u) pre = ((const u32*)(a.tdesc + g))[lane];
  const u64 tile_cols = wave_get_u64(pre, 0);
  const bool fast = !a.force_slow && tile_cols <= WGA_FAST_COL_LIMIT;
  /* granule width: 16 columns unless the tile is wider than the table covers */
  u32 gsh = a.no_table ? 8u : WGA_TBL_SHIFT;
  while ((tile_cols >> gsh) >= WGA_TBL_N) gsh++;
  const bool use_tbl = fast;
  const u32 r0 = wave_get_u32(pre, 2);
  /* op index (tile-relative) where the tile's first record ends; >= nt if it does not end here */
  const u64 re0 = wave_get_u64(pre, 12);
  const u32 kb0 = re0 < tile_end ? (u32)(re0 - tile_start) : 0xFFFFFFFFu;
  if (use_tbl)
    for (u32 k = tid; k < WGA_TBL_N + 2u; k += WGA_BLOCK) s_tbl[k] = 0u;
  if (tid < 2u) s_zero2[tid] = 0u;

  /* ---- phase A: 4 consecutive ops per thread, block scan into LDS -------------------------- */
  u32 opw[4];
  {
    u32 base = tid * 4u;
    if (base + 3 < nt) {
      u32x4_a16 v = ops_load16(a.ops + tile_start + base);
      opw[0] = v[0];
      opw[1] = v[1];
      opw[2] = v[2];
      opw[3] = v[3];
    } else {
      for (int e = 0; e < 4; e++) opw[e] = (base + e < nt) ? a.ops[tile_start + base + e] : 0u;
    }
  }
  /* exclusive (column, gap-op counts) prefix at this thread's first op, kept for record boundaries
   * beyond the first one (per-op prefix arrays would cost 8 KB of LDS and the fifth block per CU) */
  u32 my_col = 0, my_cnt = 0;
  if (fast) {
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
    /* exclusive prefixes of (columns, I bases, D bases, gap-op counts): every component stays
     * below 2^31 on the fast path */
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
      if (isi | isd) { /* ONE instance for both kinds of gap op: the body runs once per wave and op slot */
        const u32 len = opw[e] >> 4;
        const u32 slot = isi ? (x_cnt & 0xFFFFu) : (x_cnt >> 16);
        u32* const g_col = isi ? s_tg_col : s_qg_col;
        u32* const g_cum = isi ? s_tg_cum : s_qg_cum;
        g_col[slot] = x_col;
        g_cum[slot] = isi ? x_i : x_d;
        if (use_tbl) tbl_mark_event(s_tbl, x_col, len, gsh, isi ? 0u : 16u);
        x_i += isi ? len : 0u;
        x_d += isi ? 0u : len;
        x_cnt += isi ? 1u : 0x10000u;
      }
      x_col += l[e];
    }
    if (tid == WGA_BLOCK - 1) { /* sentinels: totals (two, so that index i+1 is always readable) */
      s_tot[0] = x_col;
      s_tot[1] = x_cnt;
      s_tg_col[x_cnt & 0xFFFFu] = x_col;
      s_tg_cum[x_cnt & 0xFFFFu] = x_i;
      s_tg_col[(x_cnt & 0xFFFFu) + 1u] = x_col;
      s_tg_cum[(x_cnt & 0xFFFFu) + 1u] = x_i;
      s_qg_col[x_cnt >> 16] = x_col;
      s_qg_cum[x_cnt >> 16] = x_d;
      s_qg_col[(x_cnt >> 16) + 1u] = x_col;
      s_qg_cum[(x_cnt >> 16) + 1u] = x_d;
    }
    if (use_tbl) { /* raw marks -> exclusive prefix */
      __syncthreads();
      tbl_scan(s_tbl, s_w4);
    }
  }
  __syncthreads();

  /* ---- phase B: walk the record segments of this tile ------------------------------------- */
  u32 r = r0;
  u64 cur = tile_start;
  u64 re = wave_get_u64(pre, 12);
  u32 bnd_col = 0u, bnd_ev = 0u, nseg = 0u; /* prefix at the start of the current segment */
  while (cur < tile_end) {
    while (re <= cur) {
      r++;
      re = a.op_off[r + 1];
    }
    const bool is0 = r == r0;
    const u64 rs = is0 ? wave_get_u64(pre, 10) : a.op_off[r];
    const u64 seg_end = re < tile_end ? re : tile_end;
    const u32 kb = (u32)(seg_end - tile_start);

    /* class sums of this record before the tile (only the tile's first segment can continue a
     * record; k_tile_base worked them out) and the record's geometry (k_rec_desc) */
    u64 b_mx = 0, b_i = 0, b_d = 0;
    if (rs < tile_start) { /* only the tile's first record can continue from earlier tiles */
      b_mx = wave_get_u64(pre, 4);
      b_i = wave_get_u64(pre, 6);
      b_d = wave_get_u64(pre, 8);
    }
    const u64 cb = b_mx + b_i + b_d; /* record-relative column of the segment start */
    const u64 tb = b_mx + b_d;       /* target bases consumed before it              */
    const u64 qb = b_mx + b_i;       /* query bases consumed before it               */

    /* The record's geometry stays spread over the lanes of `dsc` (layout of wga_tile_desc lanes
     * 14..31, strand in lane 3): a field costs one v_readlane where it is used instead of an
     * SGPR pair held — and spilled — across the whole segment. */
    u32 dsc = pre;
    if (!is0) {
      const u32* rp = (const u32*)(a.recs + r);
      dsc = 0u;
      if (lane >= 14u && lane < 32u) dsc = rp[lane - 14u];
      if (lane == 3u) dsc = rp[18];
    }
    const u64 t_src_len = wave_get_u64(dsc, 20), q_src_len = wave_get_u64(dsc, 24);
    u64* const bad_base = (u64*)&a.diag[r].bad_base_pos;
    u64* const panic_idx = (u64*)&a.diag[r].panic_op_idx;

    u32 col_a = 0, seg_cols = 0, icum_a = 0, dcum_a = 0;
    int ia = 0, ib = 0, ja = 0, jb = 0;
    if (fast) {
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
      col_a = bnd_col;
      seg_cols = col_b - col_a;
      const u32 eva = bnd_ev;
      bnd_col = col_b;
      bnd_ev = evb;
      ia = (int)(eva & 0xFFFFu);
      ib = (int)(evb & 0xFFFFu);
      ja = (int)(eva >> 16);
      jb = (int)(evb >> 16);
      icum_a = WGA_UNI32(s_tg_cum[ia]);
      dcum_a = WGA_UNI32(s_qg_cum[ja]);

      /* String::insert_str panics when the insertion point is beyond the string
       * (cigar.rs:507,513): an I (D) op whose target (query) consumption so far exceeds the
       * fetched slice.  Checked on the compact gap lists; the exact op index is only worked out
       * (serial rescan by the detecting thread) when that ever happens. */
      bool pan = false;
      for (int i = ia + (int)tid; i < ib; i += (int)WGA_BLOCK)
        pan |= tb + (u64)(s_tg_col[i] - col_a) - (u64)(s_tg_cum[i] - icum_a) > t_src_len;
      for (int i = ja + (int)tid; i < jb; i += (int)WGA_BLOCK)
        pan |= qb + (u64)(s_qg_col[i] - col_a) - (u64)(s_qg_cum[i] - dcum_a) > q_src_len;
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
    } else {
      /* u64 fallback for tiles wider than 2^31 columns: ops are walked serially (every thread
       * redundantly), each op's columns are written block-strided, one byte per store */
      RowSrc ts, qs;
      ts.fa = a.t_fa;
      ts.fa_bytes = a.t_fa_bytes;
      ts.src_off = wave_get_u64(dsc, 18);
      ts.src_len = t_src_len;
      ts.rc = false;
      qs.fa = a.q_fa;
      qs.fa_bytes = a.q_fa_bytes;
      qs.src_off = wave_get_u64(dsc, 22);
      qs.src_len = q_src_len;
      qs.rc = (wave_get_u32(dsc, 3) & 1u) != 0u;
      const u64 t_row_len = t_src_len + wave_get_u64(dsc, 26), q_row_len = q_src_len + wave_get_u64(dsc, 28);
      u8* const t_dst = a.out + wave_get_u64(dsc, 14);
      u8* const q_dst = a.out + wave_get_u64(dsc, 16);
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

    /* Row jobs through ONE emit_row site (keeps the kernel small enough for the I-cache):
     * 0/1 = this segment of the target / query row (rows end where a short slice ends);
     * 2/3 = once the record ends in this tile, what the slices hold beyond the CIGAR. */
    const bool rec_ends = seg_end == re;
    const u32 rec_flags = wave_get_u32(dsc, 3);
#pragma nounroll
    for (int job = 0; job < 4; job++) {
      const bool is_q = (job & 1) != 0, is_tail = job >= 2;
      if (is_tail && (!rec_ends || !(rec_flags & (is_q ? 4u : 2u)))) continue; /* no tail: nothing read, nothing computed */
      /* Ownership is static: piece p of this job goes to wave (job + 2 p + segment index) mod 4 — the two halves of
       * the two rows of a segment land on four different waves, and the rotation with the segment spreads the
       * single-piece rows of short records.  A wave that owns neither piece reads nothing of the job (rows beyond
       * WGA_SOLO_BYTES are the whole block's: only possible when the segment is that wide). */
#ifndef WGA_OWN_STRIDE
#define WGA_OWN_STRIDE 2u /* piece 1 of a job goes to the wave this many behind piece 0's */
#endif
      const u32 own0 = ((u32)job * (WGA_OWN_STRIDE == 2u ? 1u : 2u) + nseg) & 3u, own1 = (own0 + WGA_OWN_STRIDE) & 3u;
      if (!is_tail && seg_cols <= WGA_SOLO_BYTES && own0 != wave && own1 != wave) continue;
      /* this row's fields: the query ones sit 2 (offsets, gap totals) or 4 (slice) lanes after
       * the target ones. */
      const int q2 = is_q ? 2 : 0, q4 = is_q ? 4 : 0;
      const u64 gap_total = wave_get_u64(dsc, 26 + q2); /* I bases (target row) / D bases (query row) */
      const u64 L = wave_get_u64(dsc, 30);
      const u64 src_len = is_q ? q_src_len : t_src_len;
      const u64 row_len = src_len + gap_total;
#if !WGA_OWNER_SETUP
      RowSrc src;
      src.fa = is_q ? a.q_fa : a.t_fa;
      src.fa_bytes = is_q ? a.q_fa_bytes : a.t_fa_bytes;
      src.src_off = wave_get_u64(dsc, 18 + q4);
      src.src_len = src_len;
      src.rc = is_q && (wave_get_u32(dsc, 3) & 1u) != 0u;
      src.drain_min = a.drain_min;
#endif
      u64 x0, nbytes;
      if (!is_tail) {
        if (!fast) continue;
        const u64 x1 = cb + seg_cols < row_len ? cb + seg_cols : row_len;
        x0 = cb;
        nbytes = x1 > cb ? x1 - cb : 0;
      } else {
        if (!rec_ends) continue;
        x0 = L;
        nbytes = row_len > L ? row_len - L : 0;
      }
      if (nbytes == 0) continue;
      /* Rows of ordinary size are not shared out thread by thread: a row (or each half of it, cut
       * on a granule boundary) is taken by ONE wave, round-robin over the tile's row jobs — one
       * prologue and one set of queue drains per row instead of four quarter-full ones.  Only
       * rows beyond WGA_SOLO_BYTES (giant tiles) are emitted by the whole block together. */
      const bool coop = nbytes > WGA_SOLO_BYTES;
      const u32 c_row = is_tail ? 0u : col_a; /* column of the row's first byte */
      u64 cut = nbytes;                        /* first piece = [0, cut), second = [cut, nbytes) */
      if (!coop && nbytes > WGA_SPLIT_BYTES) {
        const u32 cc = (c_row + (u32)(nbytes >> 1)) & ~15u;
        if (cc > c_row && (u64)(cc - c_row) < nbytes) cut = cc - c_row;
      }
#if !WGA_OWNER_SETUP
      u8* const dst = a.out + wave_get_u64(dsc, 14 + q2) + x0;
      const u64 sb0 = is_tail ? L - gap_total : (is_q ? qb : tb);
#endif
      for (int piece = 0; piece < 2; piece++) {
        const u64 lo = piece == 0 ? 0 : cut, hi = piece == 0 ? cut : nbytes;
        if (lo >= hi) continue;
        if (!coop && (piece == 0 ? own0 : own1) != wave) continue;
#if WGA_OWNER_SETUP
        RowSrc src;
        src.fa = is_q ? a.q_fa : a.t_fa;
        src.fa_bytes = is_q ? a.q_fa_bytes : a.t_fa_bytes;
        src.src_off = wave_get_u64(dsc, 18 + q4);
        src.src_len = src_len;
        src.rc = is_q && (wave_get_u32(dsc, 3) & 1u) != 0u;
        src.drain_min = a.drain_min;
        u8* const dst = a.out + wave_get_u64(dsc, 14 + q2) + x0;
        const u64 sb0 = is_tail ? L - gap_total : (is_q ? qb : tb);
#endif
        for (u64 done = lo; done < hi; done += (1ull << 30)) {
          const u64 m = hi - done < (1ull << 30) ? hi - done : (1ull << 30);
          RowDesc rd;
          rd.c_org = is_tail ? 0u : col_a;
          rd.G_col = is_q ? s_qg_col : s_tg_col;
          rd.G_cum = is_q ? s_qg_cum : s_tg_cum;
          rd.G_adj = is_tail ? s_zero2 : rd.G_cum; /* a tail's fast path reads G_adj[0]: the gap lists are not even
                                                      initialised when the tile takes the op-serial walk */
          rd.ga = is_tail ? 0 : (is_q ? ja : ia);
          rd.gb = is_tail ? 0 : (is_q ? jb : ib);
          rd.gcum_a = is_tail ? 0u : (is_q ? dcum_a : icum_a);
          rd.sbase = is_tail ? sb0 + done : sb0;
          rd.lowmask = s_lowmask;
          rd.tbl = is_tail ? s_zero2 : s_tbl;
          rd.tsh = (is_q && !is_tail) ? 16u : 0u;
          rd.gsh = is_tail ? 31u : gsh;
          rd.queue = s_queue;
          rowsrc_prepare(src, rd.sbase);
          emit_row(dst + done, (u32)m, is_tail ? 0u : col_a + (u32)done, rd, src, coop ? tid : lane,
                   coop ? WGA_BLOCK : 64u, bad_base);
        }
      }
    }
    cur = seg_end;
    r++;
    if (cur < tile_end) re = a.op_off[r + 1];
    nseg++;
  }
}

/* v1 of the row kernel (granules stored as they are produced: lines reach the L2 in pieces): `expand_variant` 0, one block
 * per tile of the batch — and, as k_paf2maf_expand_list, the kernel of the tiles the streaming kernel (wga_kernels_k2s.h, the
 * default) leaves: records that are not clean, pool edges, tiles beyond 2^24 columns, the op-serial u64 walk beyond 2^31. */
/* Blocks go to the 8 XCDs round robin (block b runs on XCD b % 8), each with its own L2.  Every XCD gets one contiguous
 * eighth of the tiles, in order: the ~190 tiles an XCD has in flight are then neighbours — the output lines two tiles
 * share and the source windows of one record's tiles meet in one L2.  Measured (profiles/r02_k2_experiments.md):
 * 6.75 -> 6.65 ms with 2 x 50 Mb pools, 7.69 -> 6.9-7.3 ms with 2 x 1 Gb pools.  -DWGA_K2_XCD=0: tile = block. */
#ifndef WGA_K2_XCD
#define WGA_K2_XCD 1
#endif
__device__ __forceinline__ u64 xcd_tile_of_block() {
#if WGA_K2_XCD
  const u32 nt = gridDim.x, b = blockIdx.x, x = b & 7u, q = nt >> 3, r = nt & 7u;
  return (u64)(x * q + (x < r ? x : r) + (b >> 3));
#else
  return blockIdx.x;
#endif
}
__global__ __launch_bounds__(256, WGA_K2_BLOCKS) void k_paf2maf_expand(ExpandArgs a) {
  expand_tile_v1(a, xcd_tile_of_block());
}


__global__ __launch_bounds__(256, 4) void k_paf2maf_expand_list(ExpandArgs a) {
  const u32 n_list = *a.tile_count;
  for (u32 idx = blockIdx.x; idx < n_list; idx += gridDim.x) {
    expand_tile_v1(a, a.tile_list[idx]);
    __syncthreads(); /* the tile's LDS state is dead */
  }
}

/* ---- copy n variable-length snippets (MAF line text between the rows) ------------------------ */
__global__ __launch_bounds__(256) void k_scatter_bytes(u32 n, const u8* src, const u64* src_off,
                                                       u8* dst, const u64* dst_off) {
  /* one wave per snippet; snippets are tens of bytes */
  const u32 lane = threadIdx.x & 63u;
  const u64 i = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x);
  if (i >= n) return;
  const u64 s0 = src_off[i], s1 = src_off[i + 1], d0 = dst_off[i];
  for (u64 k = s0 + lane; k < s1; k += 64) dst[d0 + (k - s0)] = src[k];
}

/* ---- wga_reduce_scatter_i32: dst[i] += the sum of src[k][i] over up to WGA_PEER_MAX source arrays (n counters) ----------------
 * The sources are the other devices' buffers, read where they lie (peer access over xGMI: one load per source and thread in
 * flight, so a device's links all carry data at the same time) or, staged, copies of them in this device's scratch.  16 bytes
 * per thread and source where the pointers share their alignment; a grid-stride loop. */
#define WGA_PEER_MAX 15
struct wga_peer_srcs {
  const int* p[WGA_PEER_MAX];
};
__global__ __launch_bounds__(256) void k_add_peers_i32(int* __restrict__ dst, wga_peer_srcs srcs, int n_src, u64 n) {
  const u64 tid = (u64)blockIdx.x * 256u + threadIdx.x, stride = (u64)gridDim.x * 256u;
  bool together = true; /* dst and every source at the same offset inside their 16-byte groups */
  for (int k = 0; k < n_src; k++) together = together && ((((u64)srcs.p[k]) ^ (u64)dst) & 15ull) == 0ull;
  u64 head = together ? ((16ull - ((u64)dst & 15ull)) & 15ull) >> 2 : n; /* counters in front of the first whole group */
  if (head > n) head = n;
  const u64 groups = (n - head) >> 2, tail0 = head + (groups << 2);
  for (u64 q = tid; q < groups; q += stride) {
    const u64 i = head + (q << 2);
    u32x4_a16 acc = *(const u32x4_a16*)(dst + i);
    u32x4_a16 v[WGA_PEER_MAX];
#pragma unroll
    for (int k = 0; k < WGA_PEER_MAX; k++)
      if (k < n_src) v[k] = *(const u32x4_a16*)(srcs.p[k] + i); /* all of them requested before the first is added */
#pragma unroll
    for (int k = 0; k < WGA_PEER_MAX; k++)
      if (k < n_src) acc += v[k];
    *(u32x4_a16*)(dst + i) = acc;
  }
  for (u64 i = tid; i < n - (groups << 2); i += stride) { /* the counters outside whole groups */
    const u64 j = i < head ? i : tail0 + (i - head);
    int acc = dst[j];
    for (int k = 0; k < n_src; k++) acc += srcs.p[k][j];
    dst[j] = acc;
  }
}

#endif /* WGA_KERNELS_H */
