/*
 * wga_kernels2.h — the consumers beside paf2maf/stat:
 *   k_class_tiles          tile summaries (+ optional per-record class sums) without the stat
 *                          counters — shared first pass of pafcov and pafpseudo
 *   K5 k_pafcov_accumulate difference-array coverage marks (update_cov_vec, cigar.rs:710-741)
 *      k_cov_windows<true> marks -> per-base counts inside the replay (segmented scan per window + look-back over windows)
 *   K6 k_pafpseudo_fill    target-coordinate pseudo-MAF segments (cigar.rs:744-804)
 *   K3 k_maf_pair_stat     MAF column-pair walk (cigar.rs:298-308,344-432)
 * Same tile decomposition and helpers as wga_kernels.h.
 */
#ifndef WGA_KERNELS2_H
#define WGA_KERNELS2_H

#include "wga_kernels.h"
#include <type_traits>

/* ============================================================================================ */
/* class sums per tile (and per record)                                                         */
/* ============================================================================================ */
__global__ __launch_bounds__(256) void k_class_tiles(const u32* __restrict__ ops,
                                                     const u64* __restrict__ op_off, u32 n,
                                                     u64 n_ops, wga_tile_sum* tiles,
                                                     wga_class_sums* rec_sums) {
  const u32 lane = threadIdx.x & 63u;
  const u32 wave = WGA_WAVE_ID(threadIdx.x);
  const u64 g = (u64)blockIdx.x * 4 + wave;
  const u64 tile_start = g * WGA_TILE;
  if (tile_start >= n_ops) return;
  const u64 tile_end = tile_start + WGA_TILE < n_ops ? tile_start + WGA_TILE : n_ops;
  const u32 nt = (u32)(tile_end - tile_start);
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
      for (int e = 0; e < 4; e++) w[4 * j + e] = (base + e < nt) ? ops[tile_start + base + e] : 0u;
    }
  }
  u32 r = wga_find_rec(op_off, n, tile_start);
  const u32 r_first = r;
  u64 cur = tile_start;
  u64 tot[5] = {0, 0, 0, 0, 0}, tail[5] = {0, 0, 0, 0, 0};
  while (cur < tile_end) {
    u64 re = op_off[r + 1];
    while (re <= cur) {
      r++;
      re = op_off[r + 1];
    }
    const u64 rs = op_off[r];
    const u64 seg_end = re < tile_end ? re : tile_end;
    const u32 a = (u32)(cur - tile_start), b = (u32)(seg_end - tile_start);
    u32 s[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        u32 idx = ((u32)j * 64u + lane) * 4u + (u32)e;
        u32 op = w[4 * j + e];
        u32 cls = op_class(op & 15u);
        u32 l = (idx >= a && idx < b) ? (op >> 4) : 0u;
        s[0] += cls == CLS_MX ? l : 0u;
        s[1] += cls == CLS_I ? l : 0u;
        s[2] += cls == CLS_D ? l : 0u;
        s[3] += cls == CLS_S ? l : 0u;
        s[4] += cls == CLS_O ? l : 0u;
      }
    }
    u64 S[5];
#pragma unroll
    for (int c = 0; c < 5; c++) S[c] = wave_sum_u32_wide(s[c]); /* DPP scans on 16-bit halves: exact (a lane's sum < 2^32) */
    if (rec_sums && lane == 0) {
      u64* f = (u64*)(rec_sums + r);
      if (rs >= tile_start && re <= tile_end) {
#pragma unroll
        for (int c = 0; c < 5; c++) f[c] = S[c];
      } else {
#pragma unroll
        for (int c = 0; c < 5; c++)
          if (S[c]) atomicAdd(f + c, S[c]);
      }
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
    ts.rec = r_first;
    tiles[g] = ts;
  }
}

/* n 64-bit words from src to dst (a context-owned result handed to the caller's array) */
__global__ __launch_bounds__(256) void k_copy_u64(u64 n, const u64* __restrict__ src, u64* __restrict__ dst) {
  const u64 i = (u64)blockIdx.x * WGA_BLOCK + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

/* ============================================================================================ */
/* K5: pafcov                                                                                   */
/* ============================================================================================ */
/* Only M and = are counted; I and S do not move; every other op (D X N H P ...) moves without
 * counting (cigar.rs:720-733).  A covered span [pos, pos+len) becomes +1 at pos and -1 at pos+len
 * (both only below the target length): two marks per M/= op instead of len increments.  A wave
 * takes a tile with 16 *consecutive* ops per lane, so that a lane can walk its ops serially after
 * one wave-level exclusive scan of the position advance. */
/* Global atomics run at ~27 G/s on this part whatever their scope or locality
 * (scripts/micro/atomic_scope.hip), i.e. 19 ms for the 5e8 marks of configs[1].  So the marks are
 * not sent to memory one by one: the coverage index space is cut into windows of WGA_COV_WIN
 * counters, every (tile, record segment) piece is listed under the windows it touches
 * (k_cov_list_pieces in ONE pass over the ops: tile sums by look-back, the pieces into the tile's own slots; then a scan of the
 * window counts and k_cov_place_*), and one block per window replays its pieces with LDS
 * atomics and adds the window to memory with plain stores — it is the only writer. */
#define WGA_COV_WIN_SHIFT 13u
#define WGA_COV_WIN (1u << WGA_COV_WIN_SHIFT)

struct __attribute__((aligned(16))) wga_cov_piece {
  u32 g;       /* tile */
  u32 ab;      /* first op | end op << 16, tile-relative (<= 1024): the ops of the record segment whose marks can lie in the
                  window the piece is listed under (whole lanes of 16 ops) */
  u64 pos0;    /* coverage index (cov_off + target position) in front of op `first` */
  u64 limit;   /* coverage index one past the target's last counter */
  u32 wi;      /* window the piece is listed under */
  u32 pad;     /* WGA_COV_NARROW: the segment advances less than 2^30 bases inside its tile */
};
#define WGA_COV_NARROW 1u
/* The list pass writes a tile's first WGA_COV_TILE_CAP pieces into the tile's own slots — no atomic with an answer to wait for —
 * and further ones where WGA_COV_LISTS counters hand out places (tile g uses counter g mod WGA_COV_LISTS: one counter for all
 * tiles would take every such segment of the batch through one address), each over a region of `rcap` pieces. */
#define WGA_COV_TILE_CAP 8u
#define WGA_COV_LISTS 4096u
#define WGA_COV_READY (1ull << 63)
/* inclusive scan over the lanes of a value below 2^40 (a lane's 16 ops advance less than 16 x 2^28), and the wave's total: two
 * 32-bit DPP scans, on the low 24 bits and on the rest */
__device__ __forceinline__ u64 cov_incl_scan_u64(u64 v, u64& total) {
  const u32 lo = wave_incl_scan_u32((u32)v & 0xFFFFFFu), hi = wave_incl_scan_u32((u32)(v >> 24));
  total = (u64)wave_last_u32(lo) + ((u64)wave_last_u32(hi) << 24);
  return (u64)lo + ((u64)hi << 24);
}

/* 16 consecutive ops per lane of tile g, zeros from op nt on: every 16-byte group that starts in front of nt is loaded (the group
 * that holds the stream's last op may reach up to 12 bytes beyond it, inside the same aligned 16 bytes; the callers give what
 * it brings from there no weight — the list pass only looks at ops inside record segments) */
__device__ __forceinline__ void cov_load_ops(const u32* __restrict__ ops, u64 tile_start, u32 nt, u32 lane,
                                             u32 w[16]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u32 base = lane * 16u + (u32)j * 4u;
    u32x4_a16 v = {0u, 0u, 0u, 0u};
    if (base < nt) v = *(const u32x4_a16*)(ops + tile_start + base);
    w[4 * j + 0] = v[0];
    w[4 * j + 1] = v[1];
    w[4 * j + 2] = v[2];
    w[4 * j + 3] = v[3];
  }
}
/* an op's advance on the target: everything but I and S moves (cigar.rs:720-733) — codes 1, 4 and 9 (I, S, the rest of a split I)
 * do not */
__device__ __forceinline__ bool cov_op_moves(u32 code) { return ((0x212u >> code) & 1u) == 0u; }
#define WGA_COV_MOVES_BITS 0xFDEDu  /* bit c set: an op of code c moves on the target (the same as a mask for v_bfe_i32) */
#define WGA_COV_NOTCNT_BITS 0xFF7Eu /* bit c set: an op of code c is not counted (all but M and =) */
/* target advance of this lane's ops inside [a, b); mvl[e] = the advance of op e (16 lengths below 2^28: the sum fits 32 bits) */
__device__ __forceinline__ u64 cov_lane_moves(const u32 mvl[16], u32 lane, u32 a, u32 b) {
  u32 mv = 0;
  const u32 t = lane * 16u - a, n = b - a;
#pragma unroll
  for (int e = 0; e < 16; e++) mv += (t + (u32)e < n) ? mvl[e] : 0u;
  return (u64)mv;
}

/* Target advance of record r's ops in front of tile g (the record starts at op rs, in tile g0 = rs / WGA_TILE): the tail sums the
 * tiles g0 .. g-1 published (every one of them ends inside the record, so its last segment is the record's part of it).  Lane L
 * takes tile g-1-L (and 64 further back per round); `early` is what a first poll of round 0 — sent before the tile's other
 * segments were worked on — brought back (cov_poll_early).  Those tiles belong to blocks of the same launch with lower
 * indices, which were dispatched before this one; should one of them not have published after `spin_limit` polls (or with a
 * limit of 0: at once), the ops themselves are added up — the pass ends whatever the dispatch order is. */
__device__ __forceinline__ u64 cov_poll_early(u64* tile_tail, u64 rs, u64 g, u32 lane) {
  const u64 g0 = rs / WGA_TILE;
  return (g - g0 > (u64)lane) ? (u64)atomicAdd((unsigned long long*)&tile_tail[g - 1 - lane], 0ull) : 0ull;
}
__device__ __forceinline__ u64 cov_look_back(u64* tile_tail, const u32* __restrict__ ops, u64 rs, u64 g, u32 lane,
                                             u32 spin_limit, u64 early) {
  const u64 g0 = rs / WGA_TILE;
  u64 p = 0;
  bool gave_up = spin_limit == 0; /* wave-uniform */
  for (u64 back = 0; back < g - g0 && !gave_up; back += 64) {
    const bool mine = g - g0 > back + lane;
    const u64 k = g - 1 - back - lane;
    u64 v = back == 0 ? early : 0ull;
    u32 polls = 0;
    for (;;) {
      if (!__ballot(mine && !(v & WGA_COV_READY))) break;
      if (++polls > spin_limit) {
        gave_up = true;
        break;
      }
      if (mine && !(v & WGA_COV_READY)) v = atomicAdd((unsigned long long*)&tile_tail[k], 0ull);
    }
    p += mine ? (v & ~WGA_COV_READY) : 0ull;
  }
  if (!gave_up) return wave_sum_u32_wide((u32)p) + (wave_sum_u32_wide((u32)(p >> 32)) << 32); /* DPP, no LDS */
  u64 q = 0;
  for (u64 i = rs + lane; i < g * WGA_TILE; i += 64) {
    const u32 op = ops[i];
    q += cov_op_moves(op & 15u) ? (u64)(op >> 4) : 0ull;
  }
  return wave_sum_u64(q);
}

/* where a record stands in the coverage index space: its first base, and one past its target's last counter */
struct wga_cov_rec {
  u64 pos0, limit;
};
__global__ __launch_bounds__(256) void k_cov_rec_pos(u32 n, const u32* __restrict__ target_id, const u64* __restrict__ t_start,
                                                     const u64* __restrict__ cov_off, const u64* __restrict__ cov_len,
                                                     wga_cov_rec* __restrict__ rec_pos) {
  const u32 r = blockIdx.x * WGA_BLOCK + threadIdx.x;
  if (r >= n) return;
  const u32 tg = target_id[r];
  const u64 coff = cov_off[tg];
  wga_cov_rec rp;
  rp.pos0 = coff + t_start[r];
  rp.limit = coff + cov_len[tg];
  rec_pos[r] = rp;
}

/* what a tile's wave needs before it can start, in one load: the record of its first op and the one behind it */
struct __attribute__((aligned(16))) wga_cov_tile {
  u64 rs, re;      /* op_off[rec], op_off[rec + 1] of the record that holds the tile's first op */
  wga_cov_rec rp0; /* that record's place */
  u64 re1;         /* op_off[rec + 2] (0 without a further record) */
  wga_cov_rec rp1; /* the place of record rec + 1 */
  u32 rec, pad;
};
__global__ __launch_bounds__(256) void k_cov_tile_info(const u64* __restrict__ op_off, u32 n, u64 n_ops,
                                                       const wga_cov_rec* __restrict__ rec_pos, wga_cov_tile* __restrict__ info) {
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
  wga_cov_tile t;
  t.rec = lo;
  t.pad = 0;
  t.rs = op_off[lo];
  t.re = op_off[lo + 1];
  t.rp0 = rec_pos[lo];
  const bool more = lo + 1u < n;
  t.re1 = more ? op_off[lo + 2] : 0ull;
  t.rp1 = rec_pos[more ? lo + 1u : lo];
  info[g] = t;
}

/* The list pass: one wave per tile of 1024 ops.  Every record segment of the tile is measured (target advance per lane, scanned),
 * the tile's last segment is published for the tiles behind it, the first segment looks back for where its record stands, and
 * every (segment, window) piece is counted under its window (win_cnt) and written to the tile's slots (beyond WGA_COV_TILE_CAP
 * pieces: to the tile's list region) — k_cov_place_* take them to their windows.  Segments other
 * than the first start with their record, so only the first one waits.  The order is: publish, look back, write.  What the first two
 * segments need of their records comes with the tile's ops in one load (k_cov_tile_info), a further segment's record data
 * (op_off, k_cov_rec_pos's pair) is fetched a segment ahead.  With `rcap` = 0 the pieces beyond the slots are only counted. */
#ifndef WGA_K5_LIST_WAVES
#define WGA_K5_LIST_WAVES 5 /* waves per SIMD the register allocation aims at (5: 96 VGPRs and 12 bytes of scratch with the next tile's ops in registers; 6: 80 and 20) */
#endif
/* one tile of the list pass; w = the tile's 16 packed ops per lane */
__device__ __forceinline__ void cov_list_tile(
    const u64 g, u32 (&w)[16], const u32 lane, const u32* __restrict__ ops, const u64* __restrict__ op_off, u64 n_ops,
    const wga_cov_tile* __restrict__ tile_info, const wga_cov_rec* __restrict__ rec_pos, u64* tile_tail, u32* win_cnt,
    wga_cov_piece* tile_list, u32* tile_cnt, u64* list_cnt, wga_cov_piece* list, u64 rcap, u32 spin_limit) {
  const u64 tile_start = g * WGA_TILE;
  const u64 tile_end = tile_start + WGA_TILE < n_ops ? tile_start + WGA_TILE : n_ops;
  const u32 nt = (u32)(tile_end - tile_start);
  const wga_cov_tile tr = tile_info[g];
  const u64 rs_next = tile_end < n_ops ? tile_info[g + 1].rs : ~0ull; /* where the record of the next tile's first op starts */
#pragma unroll
  for (int e = 0; e < 16; e++) w[e] = (w[e] >> 4) & bit_mask(WGA_COV_MOVES_BITS, w[e] & 15u); /* the pass needs the ops' advance only */
  if (nt & 3u) { /* wave-uniform, the stream's last tile: what the last 16-byte group brought from behind the stream */
#pragma unroll
    for (int e = 0; e < 16; e++) w[e] = lane * 16u + (u32)e < nt ? w[e] : 0u;
  }
  const u32 region = (u32)(g % WGA_COV_LISTS);
  u64* const my_cnt = list_cnt + region;
  wga_cov_piece* const my_list = list + (u64)region * rcap;
  wga_cov_piece* const my_slots = tile_list + g * WGA_COV_TILE_CAP;
  u64 n_mine = 0; /* pieces of this tile so far (wave-uniform) */

  /* where every lane's 16 ops start and end on the target (monotone over the lanes; lanes outside [a, b) are empty) */
  auto measure = [&](u32 a, u32 b, u64& mv, u64& inc, u64& span) {
    mv = cov_lane_moves(w, lane, a, b);
    inc = cov_incl_scan_u64(mv, span);
  };
  auto publish = [&](u64 span) {
    if (lane == 0) atomicAdd((unsigned long long*)&tile_tail[g], (unsigned long long)(span | WGA_COV_READY));
  };
  auto emit = [&](const wga_cov_rec& rp, u32 a, u32 b, u64 mv, u64 inc, u64 span, u64 base) {
    const u64 pos = rp.pos0 + base; /* coverage index in front of the segment */
    if (pos >= rp.limit) return;    /* wave-uniform */
    /* marks lie in [pos, min(pos + span, limit - 1)] */
    const u64 last = pos + span < rp.limit ? pos + span : rp.limit - 1;
    const u64 wlo = pos >> WGA_COV_WIN_SHIFT, whi = last >> WGA_COV_WIN_SHIFT;
    const u64 l_end = pos + inc, l_start = l_end - mv; /* this lane's ops mark inside [l_start, l_end] */
    const u64 np = whi - wlo + 1;
    /* the pieces beyond the tile's own slots (rare: a tile of many short records, a segment across many windows) take places
     * in the tile's list region — the only atomic of the pass whose answer is waited for */
    const u64 over0 = n_mine > (u64)WGA_COV_TILE_CAP ? n_mine : (u64)WGA_COV_TILE_CAP;
    u64 place0 = 0;
    if (n_mine + np > over0) { /* wave-uniform */
      if (lane == 0) place0 = atomicAdd((unsigned long long*)my_cnt, (unsigned long long)(n_mine + np - over0));
      place0 = WGA_UNI64(__shfl(place0, 0));
    }
    for (u64 j0 = 0; j0 < np; j0 += 64) { /* a lane per window: a window replays only the lanes that can mark inside it */
      const u64 wi = wlo + j0 + lane;
      const bool on = wi <= whi;
      const u64 lo = wi << WGA_COV_WIN_SHIFT, hi = lo + (u64)(WGA_COV_WIN - 1u);
      u32 c1 = 0; /* lanes that end in front of the window (63 when all do) */
#pragma unroll
      for (u32 st = 32; st; st >>= 1) {
        const u64 v = __shfl(l_end, (int)(c1 + st - 1u));
        c1 += v < lo ? st : 0u;
      }
      u32 c2 = 0; /* lanes that start inside or in front of the window */
#pragma unroll
      for (u32 st = 32; st; st >>= 1) {
        const u64 v = __shfl(l_start, (int)(c2 + st - 1u));
        c2 += v <= hi ? st : 0u;
      }
      c2 += __shfl(l_start, (int)c2) <= hi ? 1u : 0u;
      const u32 l1 = c1, l2 = c2 ? c2 - 1u : 0u;
      const u32 a2 = a > 16u * l1 ? a : 16u * l1, b2 = b < 16u * (l2 + 1u) ? b : 16u * (l2 + 1u);
      const u64 pos_a2 = __shfl(l_start, (int)l1);
      if (on) {
        atomicAdd(&win_cnt[wi], 1u);
        wga_cov_piece pc;
        pc.g = (u32)g;
        pc.ab = a2 | (b2 << 16);
        pc.pos0 = pos_a2;
        pc.limit = rp.limit;
        pc.wi = (u32)wi;
        pc.pad = span < (1ull << 30) ? WGA_COV_NARROW : 0u; /* the replay may walk it in 32-bit window positions */
        const u64 idx = n_mine + j0 + lane;
        if (idx < (u64)WGA_COV_TILE_CAP)
          my_slots[idx] = pc;
        else if (place0 + (idx - over0) < rcap)
          my_list[place0 + (idx - over0)] = pc;
      }
    }
    n_mine += np;
  };

  const u32 r0 = tr.rec;
  const u64 rs0 = tr.rs, end0 = tr.re < tile_end ? tr.re : tile_end;
  const u32 b0 = (u32)(end0 - tile_start);
  const wga_cov_rec rp0 = tr.rp0;
  const bool waits = rs0 < tile_start; /* the first segment's record began in a tile in front */
  const u64 early = waits ? cov_poll_early(tile_tail, rs0, g, lane) : 0ull;
  /* ONE scan serves every segment of the tile: the lanes' advance summed over all 1 024 ops, whatever records they belong to.
   * The advance in front of op i (wave-uniform i) is the sum of the lanes in front of i's lane plus that lane's ops in front of
   * i: a handful of adds in every lane and two v_readlane.  A tile that advances less than 2^31 bases (NARROW: every tile of a
   * real alignment) is then handled in 32-bit positions relative to each segment's start; anything else, a segment across more
   * than the slots' windows or a tile of more than WGA_COV_TILE_CAP pieces takes the general walk below, which measures every
   * segment on its own (64-bit scans, a lane-per-window search). */
  /* w becomes the running sum inside the lane (w[e] = advance of the lane's ops 0 .. e): the advance in front of any op is then
   * one register of one lane, read with a wave-uniform register index */
#pragma unroll
  for (int e = 1; e < 16; e++) w[e] += w[e - 1]; /* 16 advances below 2^28 */
  const u32 lt = w[15];
  auto per_op_again = [&]() { /* the general walk measures ops one by one */
#pragma unroll
    for (int e = 15; e > 0; e--) w[e] -= w[e - 1];
  };
  u64 tile_total;
  const u64 P64 = cov_incl_scan_u64((u64)lt, tile_total);
  const bool narrow = tile_total < (1ull << 31); /* wave-uniform */
  const u32 Pin = (u32)P64, Pex = Pin - lt;
  auto prefix_at = [&](u32 i) -> u32 { /* NARROW only; i <= nt, wave-uniform */
    if (i >= WGA_TILE) return (u32)tile_total;
    const u32 li = i >> 4, e = WGA_UNI32(i & 15u);
    const u32 part = e ? w[(e - 1u) & 15u] : 0u;
    return wave_get_u32_dyn(Pex, li) + wave_get_u32_dyn(part, li);
  };
  if (!narrow) per_op_again();
  /* Publish first, then look back, then write: the wait for the look-back's answer stands in front of the tile's first store
   * (a wait behind stores is a wait for their acknowledgements as well — a third of the pass when it was there).  The tile's
   * last segment starts where the record of the next tile's first op starts; when that record starts with the next tile, no
   * tile behind will ask for this one's sum. */
  {
    u64 span_last = 0;
    if (rs_next < tile_end) { /* wave-uniform */
      const u32 a_last = rs_next > tile_start ? (u32)(rs_next - tile_start) : 0u;
      if (narrow) {
        span_last = tile_total - (u64)prefix_at(a_last);
      } else {
        u64 mv, inc;
        measure(a_last, nt, mv, inc, span_last);
      }
    }
    publish(span_last);
  }
  const u64 base0 = waits ? cov_look_back(tile_tail, ops, rs0, g, lane, spin_limit, early) : 0ull;
  if (narrow) {
    u32 n_p = 0; /* pieces so far; lane q keeps piece q until the walk is through (nothing is written before) */
    u32 pc_ab = 0, pc_wi = 0, pc_pad = 0;
    u64 pc_pos = 0, pc_lim = 0;
    auto seg = [&](const wga_cov_rec& rp, u32 a, u32 b, u32 Ca, u32 Cb, u64 base) -> bool {
      const u64 pos = rp.pos0 + base; /* coverage index in front of the segment */
      if (pos >= rp.limit) return true;
      const u32 span = Cb - Ca;
      const u64 last = pos + span < rp.limit ? pos + span : rp.limit - 1;
      const u64 wlo = pos >> WGA_COV_WIN_SHIFT, np64 = (last >> WGA_COV_WIN_SHIFT) - wlo + 1;
      if (np64 > (u64)(WGA_COV_TILE_CAP - n_p)) return false;
      const u32 np = (u32)np64, q0 = (u32)pos & (WGA_COV_WIN - 1u);
      /* where this lane's ops of the segment start and end, relative to the segment's start (lanes in front: 0, behind: span) */
      const u32 lo_c = Pex > Ca ? Pex : Ca, hi_c = Pin > Ca ? Pin : Ca;
      const u32 cs = (lo_c < Cb ? lo_c : Cb) - Ca, ce = (hi_c < Cb ? hi_c : Cb) - Ca;
      const u32 wf = (q0 + cs) >> WGA_COV_WIN_SHIFT, wl = (q0 + ce) >> WGA_COV_WIN_SHIFT; /* its first and last window */
      const u32 pad = span < (1u << 30) ? WGA_COV_NARROW : 0u;
      for (u32 j = 0; j < np; j++) { /* a window replays only the lanes that can mark inside it */
        u32 l1 = (u32)__popcll(__ballot(wl < j)); /* lanes that end in front of the window */
        l1 = l1 < 63u ? l1 : 63u;
        const u32 c2 = (u32)__popcll(__ballot(wf <= j)); /* lanes that start inside or in front of it */
        const u32 l2 = c2 ? c2 - 1u : 0u;
        const u32 a2 = a > 16u * l1 ? a : 16u * l1, b2 = b < 16u * (l2 + 1u) ? b : 16u * (l2 + 1u);
        const u64 pos_a2 = pos + (u64)wave_get_u32_dyn(cs, l1);
        const bool me = lane == n_p + j;
        pc_ab = me ? (a2 | (b2 << 16)) : pc_ab;
        pc_wi = me ? (u32)(wlo + j) : pc_wi;
        pc_pad = me ? pad : pc_pad;
        pc_pos = me ? pos_a2 : pc_pos;
        pc_lim = me ? rp.limit : pc_lim;
      }
      n_p += np;
      return true;
    };
    u32 Ca = 0, Cb = prefix_at(b0);
    bool ok = seg(rp0, 0u, b0, Ca, Cb, base0);
    u64 cur = end0;
    u32 r = r0 + 1;
    u64 re_next = tr.re1;
    wga_cov_rec rp_next = tr.rp1;
    while (ok && cur < tile_end) { /* the segments behind the first one start with their records */
      u64 re = re_next;
      wga_cov_rec rp = rp_next;
      while (re <= cur) { /* records without ops */
        r++;
        re = op_off[r + 1];
        rp = rec_pos[r];
      }
      const u64 seg_end = re < tile_end ? re : tile_end;
      const u32 a = (u32)(cur - tile_start), b = (u32)(seg_end - tile_start);
      if (seg_end < tile_end) { /* the next record's end and place travel behind this segment's work */
        re_next = op_off[r + 2];
        rp_next = rec_pos[r + 1];
      }
      Ca = Cb;
      Cb = prefix_at(b);
      ok = seg(rp, a, b, Ca, Cb, 0ull);
      cur = seg_end;
      r++;
    }
    if (ok) {
      if (lane < n_p) {
        wga_cov_piece pc;
        pc.g = (u32)g;
        pc.ab = pc_ab;
        pc.pos0 = pc_pos;
        pc.limit = pc_lim;
        pc.wi = pc_wi;
        pc.pad = pc_pad;
        atomicAdd(&win_cnt[pc_wi], 1u);
        my_slots[lane] = pc;
      }
      if (lane == 0) tile_cnt[g] = n_p;
      return;
    }
    per_op_again();
  }
  {
    u64 mv, inc, span;
    measure(0u, b0, mv, inc, span);
    emit(rp0, 0u, b0, mv, inc, span, base0);
  }
  u64 cur = end0;
  u32 r = r0 + 1;
  u64 re_next = tr.re1; /* another record follows in this tile when end0 < tile_end */
  wga_cov_rec rp_next = tr.rp1;
  while (cur < tile_end) { /* the segments behind the first one start with their records */
    u64 re = re_next;
    wga_cov_rec rp = rp_next;
    while (re <= cur) { /* records without ops */
      r++;
      re = op_off[r + 1];
      rp = rec_pos[r];
    }
    const u64 seg_end = re < tile_end ? re : tile_end;
    const u32 a = (u32)(cur - tile_start), b = (u32)(seg_end - tile_start);
    if (seg_end < tile_end) { /* the next record's end and place travel behind this segment's work */
      re_next = op_off[r + 2];
      rp_next = rec_pos[r + 1];
    }
    u64 mv, inc, span;
    measure(a, b, mv, inc, span);
    emit(rp, a, b, mv, inc, span, 0ull);
    cur = seg_end;
    r++;
  }
  if (lane == 0) tile_cnt[g] = n_mine < (u64)WGA_COV_TILE_CAP ? (u32)n_mine : WGA_COV_TILE_CAP;
}

/* A grid of RESIDENT waves (the host sizes it by the occupancy the runtime reports): wave j of W takes the tiles j, W + j,
 * 2 W + j ... and requests the ops of its next tile before it works on the current one.  What bounded the pass with one tile
 * per wave was the time a wave spends waiting for its 4 KB of ops with nothing else to do (7 waves per SIMD x 4 KB in flight),
 * not instructions and not HBM.  A tile looks back at the tile in front of it, which belongs to the wave in front in the same
 * round (wave 0: to the last wave's round before) and is published at the start of that wave's round: the waves move through
 * the rounds side by side.  Should a wave it waits for not be running (a grid larger than what is resident, the emulator's one
 * block at a time), the look-back's poll limit ends the wait and the wave adds up the ops itself. */
__global__ __launch_bounds__(256, WGA_K5_LIST_WAVES) void k_cov_list_pieces(
    const u32* __restrict__ ops, const u64* __restrict__ op_off, u64 n_ops, const wga_cov_tile* __restrict__ tile_info,
    const wga_cov_rec* __restrict__ rec_pos, u64* tile_tail, u32* win_cnt, wga_cov_piece* tile_list, u32* tile_cnt, u64* list_cnt,
    wga_cov_piece* list, u64 rcap, u32 spin_limit) {
  const u32 lane = threadIdx.x & 63u;
  const u64 W = (u64)gridDim.x * 4;
  u64 g = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x);
  if (g * WGA_TILE >= n_ops) return;
  auto tile_ops = [&](u64 k) { return k * WGA_TILE + WGA_TILE < n_ops ? WGA_TILE : (u32)(n_ops - k * WGA_TILE); };
  u32 wn[16];
  cov_load_ops(ops, g * WGA_TILE, tile_ops(g), lane, wn);
  for (; g * WGA_TILE < n_ops; g += W) { /* wave-uniform */
    u32 w[16];
#pragma unroll
    for (int e = 0; e < 16; e++) w[e] = wn[e];
    if ((g + W) * WGA_TILE < n_ops) cov_load_ops(ops, (g + W) * WGA_TILE, tile_ops(g + W), lane, wn);
    cov_list_tile(g, w, lane, ops, op_off, n_ops, tile_info, rec_pos, tile_tail, win_cnt, tile_list, tile_cnt, list_cnt, list, rcap,
                  spin_limit);
  }
}

/* the listed pieces go to their windows: a piece takes the next place of its window (win_fill, zero before) — an atomic with an
 * answer per piece, but of threads that have nothing else to wait for */
__global__ __launch_bounds__(256) void k_cov_place_tiles(u64 n_tiles, const u32* __restrict__ tile_cnt,
                                                         const wga_cov_piece* __restrict__ tile_list, u32* win_fill,
                                                         const u64* __restrict__ win_off, wga_cov_piece* pieces) {
  const u64 t = (u64)blockIdx.x * WGA_BLOCK + threadIdx.x;
  const u64 g = t / WGA_COV_TILE_CAP;
  if (g >= n_tiles || (u32)(t % WGA_COV_TILE_CAP) >= tile_cnt[g]) return;
  const wga_cov_piece pc = tile_list[t];
  pieces[win_off[pc.wi] + atomicAdd(&win_fill[pc.wi], 1u)] = pc;
}
__global__ __launch_bounds__(256) void k_cov_place_pieces(const u64* __restrict__ list_cnt, const wga_cov_piece* __restrict__ list,
                                                          u64 rcap, u32* win_fill, const u64* __restrict__ win_off,
                                                          wga_cov_piece* pieces) {
  const u32 region = blockIdx.y;
  const u64 i = (u64)blockIdx.x * WGA_BLOCK + threadIdx.x;
  if (i >= list_cnt[region]) return;
  const wga_cov_piece pc = list[(u64)region * rcap + i];
  pieces[win_off[pc.wi] + atomicAdd(&win_fill[pc.wi], 1u)] = pc;
}

struct ScanU32 {
  const u32* in;
  __device__ u64 operator()(u32 i) const { return (u64)in[i]; }
};

/* One block per window.  What bounds the replay is the instructions a CU can issue (about one vector instruction per cycle; the
 * walk below spends ~75 per 256-op step where the 64-bit form spent ~210) and then the bytes it has in flight: a piece is a
 * stretch of ops at a random place of the op stream, behind a descriptor — two dependent loads of ≈ 2 us each under load.  So a
 * wave keeps the first ops of WGA_COV_AHEAD pieces in flight behind the one it works on and their descriptors as far again
 * ahead of those, a block has WGA_COV_WAVES waves, and the window's counters themselves are read before the replay starts
 * instead of after it.  Lanes behind a piece's last op do not load.
 *
 * A piece the list pass marked NARROW (its segment advances less than 2^30 bases inside its tile) is walked in 32-bit positions
 * relative to the window: a mark's counter index is `r | not_counted` (all ones for an op that is not M / =), and one unsigned
 * compare with Lc = min(limit - w0, WGA_COV_WIN) says whether the mark lies in the window and below the target's length.
 *
 * FINAL (wga_pafcov_finalize, wga_pafcov_accumulate_final): the block goes on to turn marks into counts — the window's counters
 * (what the array held + this call's marks) are scanned in LDS with the restarts the target ranges ask for, the window publishes
 * its sum (or, when a range starts or ends inside it, what it hands on) and looks back over the windows in front (decoupled
 * look-back: window i waits for the SUMS of the windows in front of it, which they publish as soon as their own replay is
 * done, not for their look-backs), and writes counts.  `rng_lo` / `rng_hi` are the targets' [first, one past last] counter
 * indices in ascending order, disjoint (the host sorts them). */
#define WGA_COV_WAVES 8u
#define WGA_COV_BLOCK (64u * WGA_COV_WAVES)
#ifndef WGA_COV_AHEAD
#define WGA_COV_AHEAD 2
#endif
#define WGA_COVF_AGG (1ull << 62)
#define WGA_COVF_PREFIX (2ull << 62)

/* ops [i0, i0 + 4) of a tile for this lane (i0 a multiple of 4), zeros from `lim` on: the 16-byte group is loaded when it starts
 * in front of `lim` — a group that holds the stream's last op may reach up to 12 bytes beyond it, inside the same aligned
 * 16 bytes (never another page), and what it brings from there lies outside every piece.  One conditional load into zeroed
 * registers: the compiler needs no moves behind it, so the load of a further step really travels behind this step's work. */
__device__ __forceinline__ void cov_load4(const u32* __restrict__ ops, u64 tile_start, u32 lim, u32 i0, u32 (&w)[4]) {
  u32x4_a16 v = {0u, 0u, 0u, 0u};
  if (i0 < lim) v = *(const u32x4_a16*)(ops + tile_start + i0);
  w[0] = v[0], w[1] = v[1], w[2] = v[2], w[3] = v[3];
}
/* where a piece's loads end: its last op rounded up to whole 16-byte groups, inside the tile */
__device__ __forceinline__ u32 cov_piece_lim(u64 n_ops, const wga_cov_piece& pc) {
  const u64 tile_start = (u64)pc.g * WGA_TILE;
  const u32 nt = tile_start + WGA_TILE < n_ops ? WGA_TILE : (u32)(n_ops - tile_start);
  const u32 b4 = ((pc.ab >> 16) + 3u) & ~3u;
  return b4 < nt ? b4 : nt;
}
/* the four ops of this lane in the first 256-op step of a piece */
__device__ __forceinline__ void cov_step_ops(const u32* __restrict__ ops, u64 n_ops, const wga_cov_piece& pc, u32 lane,
                                             u32 (&w)[4]) {
  cov_load4(ops, (u64)pc.g * WGA_TILE, cov_piece_lim(n_ops, pc), ((pc.ab & 0xFFFFu) & ~3u) + lane * 4u, w);
}

/* the last range that starts at or in front of counter k (n when none does) */
__device__ __forceinline__ u32 cov_find_range(const u64* __restrict__ rng_lo, u32 n, u64 k) {
  if (n == 0u || rng_lo[0] > k) return n;
  u32 lo = 0, hi = n;
  while (hi - lo > 1u) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (rng_lo[mid] <= k)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

/* what the windows in front of window wi hand on: the sums of the windows back to the nearest one that published a PREFIX, and
 * that prefix (wave-wide; every lane returns the value).  Window 0 always publishes a PREFIX. */
__device__ __forceinline__ u32 cov_windows_in_front(u64* win_state, u64 wi, u32 lane) {
  u32 acc = 0;
  for (u64 back = 0;; back += 64) {
    const bool mine = wi > back + lane;
    const u64 k = wi - 1 - back - lane;
    u64 v = 0;
    u64 pref;
    for (u32 polls = 0;; polls++) {
      if (mine && (v >> 62) == 0ull) v = atomicAdd((unsigned long long*)&win_state[k], 0ull);
      pref = __ballot(mine && (v >> 62) == 2ull);
      const u64 empty = __ballot(mine && (v >> 62) == 0ull);
      const u64 front = pref ? ((pref & (0ull - pref)) - 1ull) : ~0ull; /* the lanes nearer than the nearest prefix */
      if (!(empty & front)) break;
      if (polls) WGA_SLEEP(8); /* the windows waited for are still replaying: do not crowd their loads */
    }
    const u32 first = pref ? (u32)__ffsll((unsigned long long)pref) - 1u : 64u;
    acc += wave_sum_u32((mine && lane <= first) ? (u32)v : 0u);
    if (pref) return acc;
  }
}

template <bool FINAL>
__global__ __launch_bounds__(WGA_COV_BLOCK, 8) void k_cov_windows(const u32* __restrict__ ops, u64 n_ops,
                                                               const wga_cov_piece* __restrict__ pieces,
                                                               const u64* __restrict__ win_off, int* cov, u64 n_cov,
                                                               const u64* __restrict__ rng_lo, const u64* __restrict__ rng_hi,
                                                               u32 n_rng, u64* win_state, const u32* __restrict__ order) {
  __shared__ int s_win[WGA_COV_WIN];
  __shared__ u32 s_ws[WGA_COV_WAVES + 1];
  __shared__ u32 s_wf[WGA_COV_WAVES];
  constexpr int D = WGA_COV_AHEAD;
  constexpr u32 PER = WGA_COV_WIN / WGA_COV_BLOCK;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  /* FINAL: the blocks take the windows in `order` — the windows that start a range (or lie outside every range) first, then every
   * range's second window, third ... — so that the ~1 000 windows in flight at one time are a few consecutive ones of MANY
   * ranges instead of a thousand consecutive ones of one: a window only waits for the windows of its own range in front of it,
   * and those were dispatched long before (the host builds the order; the window a block waits for always has a lower rank) */
  const u64 wi = (FINAL && order) ? (u64)order[blockIdx.x] : (u64)blockIdx.x;
  const u64 p_lo = win_off ? win_off[wi] : 0ull, p_hi = win_off ? win_off[wi + 1] : 0ull;
  if (!FINAL && p_lo == p_hi) return; /* block-uniform */
  const u64 w0 = wi << WGA_COV_WIN_SHIFT;
  int old[PER]; /* this block is the window's only writer */
#pragma unroll
  for (u32 j = 0; j < PER; j++) {
    const u64 k = w0 + tid + j * WGA_COV_BLOCK;
    old[j] = k < n_cov ? cov[k] : 0;
  }
  /* the marks are added to zeros (and only counters with a mark are written) or, when counts are made, to what the array held */
#pragma unroll
  for (u32 j = 0; j < PER; j++) s_win[tid + j * WGA_COV_BLOCK] = FINAL ? old[j] : 0;
  __syncthreads();
  /* a wave's pieces one after the other: dq[k] / wq[k] = descriptor / first ops of the piece k rounds behind the current one */
  const u64 p0 = p_lo + wave;
  if (p0 < p_hi) { /* wave-uniform */
    wga_cov_piece dq[2 * D + 1];
    u32 wq[D + 1][4];
#pragma unroll
    for (int k = 0; k <= 2 * D; k++) {
      const u64 q = p0 + (u64)k * WGA_COV_WAVES;
      dq[k] = pieces[q < p_hi ? q : p_lo];
    }
#pragma unroll
    for (int k = 0; k <= D; k++) {
      wq[k][0] = wq[k][1] = wq[k][2] = wq[k][3] = 0u;
      if (p0 + (u64)k * WGA_COV_WAVES < p_hi) cov_step_ops(ops, n_ops, dq[k], lane, wq[k]);
    }
    for (u64 p = p0; p < p_hi; p += WGA_COV_WAVES) {
      const wga_cov_piece pc = dq[0];
      u32 w[4] = {wq[0][0], wq[0][1], wq[0][2], wq[0][3]};
#pragma unroll
      for (int k = 0; k < 2 * D; k++) dq[k] = dq[k + 1];
#pragma unroll
      for (int k = 0; k < D; k++) {
#pragma unroll
        for (int e = 0; e < 4; e++) wq[k][e] = wq[k + 1][e];
      }
      if (p + (u64)(2 * D + 1) * WGA_COV_WAVES < p_hi) dq[2 * D] = pieces[p + (u64)(2 * D + 1) * WGA_COV_WAVES];
      if (p + (u64)(D + 1) * WGA_COV_WAVES < p_hi) cov_step_ops(ops, n_ops, dq[D], lane, wq[D]);
      const u64 tile_start = (u64)pc.g * WGA_TILE;
      const u32 lim = cov_piece_lim(n_ops, pc);
      const u32 a = pc.ab & 0xFFFFu, b = pc.ab >> 16;
      if (pc.pad & WGA_COV_NARROW) { /* wave-uniform */
        const u64 room = pc.limit - w0; /* the piece is listed under this window: limit > w0 */
        const u32 Lc = room < (u64)WGA_COV_WIN ? (u32)room : WGA_COV_WIN;
        u32 rb = (u32)(pc.pos0 - w0); /* two's complement: a position in front of the window compares above Lc */
        for (u32 s0 = a & ~3u; s0 < b; s0 += 256u) {
          const u32 i0 = s0 + lane * 4u;
          const bool more = s0 + 256u < b; /* wave-uniform: a further step's ops travel behind this one's work */
          u32 wn[4] = {0u, 0u, 0u, 0u};
          if (more) cov_load4(ops, tile_start, lim, i0 + 256u, wn);
          if (s0 < a || s0 + 256u > b) { /* wave-uniform: ops outside the piece become an I of no bases */
#pragma unroll
            for (int e = 0; e < 4; e++) w[e] = (i0 + (u32)e - a < b - a) ? w[e] : 1u;
          }
          u32 lm[4], mv = 0;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            lm[e] = (w[e] >> 4) & bit_mask(WGA_COV_MOVES_BITS, w[e] & 15u);
            mv += lm[e];
          }
          const u32 inc = wave_incl_scan_u32(mv);
          u32 r = rb + (inc - mv);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const u32 nc = bit_mask(WGA_COV_NOTCNT_BITS, w[e] & 15u);
            const u32 ku = r | nc, kd = (r + (w[e] >> 4)) | nc;
            if (ku < Lc) atomicAdd(&s_win[ku], 1);
            if (kd < Lc) atomicAdd(&s_win[kd], -1);
            r += lm[e];
          }
          rb += wave_last_u32(inc);
#pragma unroll
          for (int e = 0; e < 4; e++) w[e] = wn[e];
        }
      } else {
        u64 pos_base = pc.pos0;
        /* the piece's ops, 4 consecutive ones per lane and 256 per step (a is a multiple of 16 or the segment's first op) */
        for (u32 s0 = a & ~3u; s0 < b; s0 += 256u) {
          const u32 i0 = s0 + lane * 4u;
          const bool more = s0 + 256u < b;
          u32 wn[4] = {0u, 0u, 0u, 0u};
          if (more) cov_load4(ops, tile_start, lim, i0 + 256u, wn);
          u32 mv32 = 0; /* four lengths below 2^28 */
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const u32 idx = i0 + (u32)e;
            mv32 += (idx - a < b - a && cov_op_moves(w[e] & 15u)) ? w[e] >> 4 : 0u;
          }
          const u64 mv = mv32;
          u64 step_moves;
          const u64 inc = cov_incl_scan_u64(mv, step_moves);
          u64 pos = pos_base + (inc - mv);
#pragma unroll
          for (int e = 0; e < 4; e++) { /* the conditions first, then one branch per mark */
            const u32 idx = i0 + (u32)e;
            const u32 code = w[e] & 15u;
            const u64 len = w[e] >> 4;
            const bool in = idx - a < b - a;
            const bool counts = in && (code == WGA_OP_M || code == WGA_OP_EQ) && pos < pc.limit;
            const u64 pe = pos + len;
            const bool up = counts && pos - w0 < (u64)WGA_COV_WIN;
            const bool down = counts && pe < pc.limit && pe - w0 < (u64)WGA_COV_WIN;
            if (up) atomicAdd(&s_win[(u32)(pos - w0)], 1);
            if (down) atomicAdd(&s_win[(u32)(pe - w0)], -1);
            pos += (in && cov_op_moves(code)) ? len : 0ull;
          }
          pos_base += step_moves;
#pragma unroll
          for (int e = 0; e < 4; e++) w[e] = wn[e];
        }
      }
    }
  }
  __syncthreads();
  if (!FINAL) {
    /* the window goes to memory; counters without a mark are not touched */
#pragma unroll
    for (u32 j = 0; j < PER; j++) {
      const int v = s_win[tid + j * WGA_COV_BLOCK];
      if (v) cov[w0 + tid + j * WGA_COV_BLOCK] = old[j] + v;
    }
    return;
  }
  /* ---- marks -> counts ---- */
  /* where the window lies among the target ranges (block-uniform) */
  const u32 t0 = cov_find_range(rng_lo, n_rng, w0);
  const u64 t0_lo = t0 < n_rng ? rng_lo[t0] : 0ull, t0_hi = t0 < n_rng ? rng_hi[t0] : 0ull;
  const bool need_carry = t0_lo < w0 && w0 < t0_hi;                        /* the first counter goes on inside a range */
  const bool plain = w0 < t0_hi && w0 + (u64)WGA_COV_WIN <= t0_hi;         /* the whole window lies inside one range */
  /* a thread's PER consecutive counters: sums since the last restart, `lead` = counters in front of the thread's first restart.
   * A plain window keeps them in registers; a window with range borders walks them in LDS, one at a time (rare, and its
   * bookkeeping would otherwise cost every window registers) */
  u32 v[PER];
  const u32 c0 = tid * PER;
  u32 run = 0, lead = PER, inmask = 0;
  bool restart = false;
  if (plain) {
#pragma unroll
    for (u32 e = 0; e < PER; e++) {
      run += (u32)s_win[c0 + e];
      v[e] = run;
    }
  } else {
    const u64 k0 = w0 + c0;
    u32 t = cov_find_range(rng_lo, n_rng, k0);
    u64 lo = t < n_rng ? rng_lo[t] : 0ull, hi = t < n_rng ? rng_hi[t] : 0ull;
    u64 nxt = t < n_rng ? (t + 1u < n_rng ? rng_lo[t + 1u] : ~0ull) : (n_rng ? rng_lo[0] : ~0ull);
#pragma unroll 1
    for (u32 e = 0; e < PER; e++) {
      const u64 k = k0 + e;
      while (k >= nxt) { /* the next range starts here (the host leaves out ranges of no counters) */
        t = t < n_rng ? t + 1u : 0u;
        lo = rng_lo[t];
        hi = rng_hi[t];
        nxt = t + 1u < n_rng ? rng_lo[t + 1u] : ~0ull;
      }
      const bool inside = t < n_rng && k >= lo && k < hi;
      if (!inside || k == lo) {
        if (!restart) lead = e;
        restart = true;
        run = 0;
      }
      if (inside) {
        run += (u32)s_win[c0 + e];
        s_win[c0 + e] = (int)run;
        inmask |= 1u << e;
      }
    }
  }
  /* segmented scan over the block's threads: (sum since the last restart, a restart seen) */
  u32 s = run;
  int f = restart ? 1 : 0;
#pragma unroll
  for (u32 d = 1; d < 64u; d <<= 1) {
    const u32 s2 = __shfl_up(s, d);
    const int f2 = __shfl_up(f, d);
    if (lane >= d) {
      if (!f) s += s2;
      f |= f2;
    }
  }
  u32 es = __shfl_up(s, 1u); /* what the lanes in front hand to this one */
  int ef = __shfl_up(f, 1u);
  if (lane == 0u) es = 0u, ef = 0;
  if (lane == 63u) s_ws[wave] = s, s_wf[wave] = (u32)f;
  __syncthreads();
  u32 cs = 0, cf = 0, Ws = 0, Wf = 0; /* the waves in front of this one; the whole window */
#pragma unroll
  for (u32 k = 0; k < WGA_COV_WAVES; k++) {
    const u32 a_s = s_ws[k], a_f = s_wf[k];
    if (k == wave) cs = Ws, cf = Wf;
    Ws = a_f ? a_s : Ws + a_s;
    Wf |= a_f;
  }
  const u32 ps = ef ? es : cs + es; /* the threads in front of this one */
  const u32 pf = cf | (u32)ef;
  if (wave == 0u) {
    /* a window with a restart hands on what stands behind its last one, whatever comes in; so does one that needs nothing */
    const bool final_now = Wf || !need_carry;
    if (lane == 0u)
      atomicMax((unsigned long long*)&win_state[wi], (unsigned long long)((final_now ? WGA_COVF_PREFIX : WGA_COVF_AGG) | (u64)Ws));
    u32 cin = 0;
    if (need_carry) {
      cin = cov_windows_in_front(win_state, wi, lane);
      if (!final_now && lane == 0u)
        atomicMax((unsigned long long*)&win_state[wi], (unsigned long long)(WGA_COVF_PREFIX | (u64)(u32)(cin + Ws)));
    }
    if (lane == 0u) s_ws[WGA_COV_WAVES] = cin;
  }
  __syncthreads();
  const u32 add = ps + (pf ? 0u : s_ws[WGA_COV_WAVES]);
  if (plain) { /* whole lines through LDS */
#pragma unroll
    for (u32 e = 0; e < PER; e++) s_win[c0 + e] = (int)(v[e] + add);
    __syncthreads();
#pragma unroll
    for (u32 j = 0; j < PER; j++) cov[w0 + tid + j * WGA_COV_BLOCK] = s_win[tid + j * WGA_COV_BLOCK];
  } else { /* counters between the ranges stay as they are */
#pragma unroll 1
    for (u32 e = 0; e < PER; e++)
      if ((inmask >> e) & 1u) cov[w0 + c0 + e] = (int)((u32)s_win[c0 + e] + (e < lead ? add : 0u));
  }
}

/* ============================================================================================ */
/* K6: pafpseudo                                                                                */
/* ============================================================================================ */
struct PseudoArgs {
  const u32* ops;
  const u64* op_off;
  const u8* strand_neg;
  u32 n;
  u64 n_ops;
  const wga_tile_sum* tiles;
  const wga_class_sums* rec_sums;
  int base_mode;
  const u8* q_fa;
  u64 q_fa_bytes;
  const u64* q_src_off;
  const u64* q_src_len;
  const u64* skip;
  u8* out;
  const u64* dst_off;
  wga_rec_diag* diag;
  const u32* tile_count; /* k_pafpseudo_fill_list: the blocks loop over tile_list[0 .. *tile_count) */
  const u32* tile_list;
};

/* symbol mode: '1' for M/=, '0' for X, '-' for D, nothing for the rest (cigar.rs:760-796) */
__device__ __forceinline__ u32 pseudo_symbol(u32 code) {
  return (code == WGA_OP_M || code == WGA_OP_EQ)
             ? 0x31313131u
             : code == WGA_OP_X ? 0x30303030u
                                : (code == WGA_OP_D || code == WGA_OP_D_CONT) ? 0x2D2D2D2Du : 0u;
}

/* fill N bytes whose value depends only on the op covering the column (symbol mode) */
__device__ __forceinline__ void emit_symbols(u8* dst, u32 N, u32 c0, const u32* s_col,
                                             const u32* s_sym, int ka, int kb, const u32* optbl, u32 gsh,
                                             const u32x4_a16* lowmask, u32 tid, u32 nthreads) {
  if (N == 0) return;
  const u64 A = (u64)dst, E = A + N;
  const u64 first = A >> 4, last = (E - 1) >> 4;
  for (u64 ch = first + tid; ch <= last; ch += nthreads) {
    const u64 base_addr = ch << 4;
    const u32 a0 = base_addr < A ? (u32)(A - base_addr) : 0u;
    const u32 b0 = base_addr + 16 > E ? (u32)(E - base_addr) : 16u;
    const u32 cz = c0 + (u32)(base_addr - A);
    u32 c = cz + a0;
    const u32 c_end = cz + b0;
    /* an op that starts at or before c: the last one that starts before c's granule (optbl = ops that start
     * before each granule), or the segment's first; ops that end before c are stepped over below */
    (void)kb;
    int k = (int)optbl[c >> gsh] - 1; /* c, not cz: cz wraps below zero for a row that starts mid-granule */
    k = k < ka ? ka : k;
    u32 o[4] = {0u, 0u, 0u, 0u};
    while (c < c_end) {
      u32 oe = s_col[k + 1];
      u32 pe = oe < c_end ? oe : c_end;
      if (pe > c) {
        const u32 sym = s_sym[k];
        const u32 W[4] = {sym, sym, sym, sym};
        merge16(o, W, (int)(c - cz), (int)(pe - cz), lowmask);
        c = pe;
      }
      k++;
    }
    if (a0 == 0u && b0 == 16u) {
      u32x4_a16 v = {o[0], o[1], o[2], o[3]};
      *(u32x4_a16*)base_addr = v;
    } else {
      u8* p = (u8*)base_addr;
      for (u32 j = a0; j < b0; j++) {
        u32 d = j >> 2;
        u32 word = d == 0 ? o[0] : d == 1 ? o[1] : d == 2 ? o[2] : o[3];
        p[j] = (u8)(word >> (8u * (j & 3u)));
      }
    }
  }
}

/* BASE = base mode (query bases; the row emitter of K2) or symbol mode: two kernels, so that the symbol one does
 * not carry the emitter's registers and LDS */
#ifndef WGA_K6_BLOCKS_BASE
#define WGA_K6_BLOCKS_BASE 4
#endif
#ifndef WGA_K6_BLOCKS_SYM
#define WGA_K6_BLOCKS_SYM 6
#endif
template <bool BASE>
__device__ __forceinline__ void pseudo_tile(const PseudoArgs& a, const u64 g) {
  /* the event lists and the chunk queue belong to the row emitter (base mode); symbol mode keeps 13 KB of LDS */
  __shared__ u32 s_col[WGA_TILE + 1];                  /* exclusive prefix of target columns (M = X D)          */
  __shared__ u32 s_ev[WGA_TILE + 1];                   /* exclusive count of event ops (D, I, S)                */
  __shared__ u32 s_sym[BASE ? 1 : WGA_TILE + 1];       /* symbol-mode byte of the op                            */
  __shared__ u32 s_g_col[BASE ? WGA_TILE + 2 : 2];     /* events: column                                        */
  __shared__ u32 s_g_cum[BASE ? WGA_TILE + 2 : 2];     /*         '-' bases before (D)                          */
  __shared__ u32 s_g_adj[BASE ? WGA_TILE + 2 : 2];     /*         D bases - (I+S) bases before (wrapping)       */
  __shared__ u32 s_tbl[WGA_TBL_N + 2];                 /* events that start before each column granule          */
  __shared__ u32 s_zero2[2];
  __shared__ u64 s_w[5];
  __shared__ u32 s_w4[4];
  __shared__ u64 s_red[4][4];
  __shared__ u32x4_a16 s_lowmask[17];
  __shared__ u32 s_queue[BASE ? 4 * WGA_QCAP : 4];

  const u32 tid = threadIdx.x;
  build_lowmask(s_lowmask);
  const u32 lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  const u64 tile_start = g * WGA_TILE;
  const u64 tile_end = tile_start + WGA_TILE < a.n_ops ? tile_start + WGA_TILE : a.n_ops;
  const u32 nt = (u32)(tile_end - tile_start);
  const wga_tile_sum tsum = a.tiles[g];
  const bool fast = tsum.tot[CLS_MX] + tsum.tot[CLS_D] + tsum.tot[CLS_I] + tsum.tot[CLS_S] <=
                    WGA_FAST_COL_LIMIT;
  u32 gsh = WGA_TBL_SHIFT;
  while (((tsum.tot[CLS_MX] + tsum.tot[CLS_D]) >> gsh) >= WGA_TBL_N) gsh++;
  if (fast)
    for (u32 k = tid; k < WGA_TBL_N + 2u; k += WGA_BLOCK) s_tbl[k] = 0u;
  if (tid < 2u) s_zero2[tid] = 0u;

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
  if (fast) {
    u32 cls[4];
    u32 l[4], sl = 0, sd = 0, sis = 0, cnt = 0;
    for (int e = 0; e < 4; e++) {
      u32 len = opw[e] >> 4;
      cls[e] = op_class(opw[e] & 15u);
      l[e] = (cls[e] == CLS_MX || cls[e] == CLS_D) ? len : 0u;
      sl += l[e];
      sd += cls[e] == CLS_D ? len : 0u;
      sis += (cls[e] == CLS_I || cls[e] == CLS_S) ? len : 0u;
      cnt += (cls[e] == CLS_D || cls[e] == CLS_I || cls[e] == CLS_S) ? 1u : 0u;
    }
    u64 totA, totB;
    u64 exA = block_excl_scan_u64((u64)sl | ((u64)sd << 32), s_w, &totA);
    u64 exB = block_excl_scan_u64((u64)sis | ((u64)cnt << 32), s_w, &totB);
    u32 x_col = (u32)exA, x_d = (u32)(exA >> 32), x_is = (u32)exB, x_cnt = (u32)(exB >> 32);
    for (int e = 0; e < 4; e++) {
      u32 k = tid * 4u + (u32)e;
      s_col[k] = x_col;
      s_ev[k] = x_cnt;
      if (!BASE) {
        s_sym[k] = pseudo_symbol(opw[e] & 15u);
        if (k < nt) atomicAdd(&s_tbl[x_col >> gsh], 1u); /* symbol mode: the table counts op starts per granule */
      }
      if (cls[e] == CLS_D || cls[e] == CLS_I || cls[e] == CLS_S) {
        if (BASE) {
          s_g_col[x_cnt] = x_col;
          s_g_cum[x_cnt] = x_d;
          s_g_adj[x_cnt] = x_d - x_is;
          tbl_mark_event(s_tbl, x_col, cls[e] == CLS_D ? (opw[e] >> 4) : 0u, gsh, 0u);
        }
        if (cls[e] == CLS_D)
          x_d += opw[e] >> 4;
        else
          x_is += opw[e] >> 4;
        x_cnt += 1u;
      }
      x_col += l[e];
    }
    if (tid == WGA_BLOCK - 1) {
      s_col[WGA_TILE] = x_col;
      s_ev[WGA_TILE] = x_cnt;
      if (BASE) {
        s_g_col[x_cnt] = s_g_col[x_cnt + 1u] = x_col;
        s_g_cum[x_cnt] = s_g_cum[x_cnt + 1u] = x_d;
        s_g_adj[x_cnt] = s_g_adj[x_cnt + 1u] = x_d - x_is;
      }
    }
    __syncthreads(); /* raw marks -> exclusive prefix */
    tbl_scan(s_tbl, s_w4);
  }
  __syncthreads();

  u32 r = (u32)tsum.rec;
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

    u64 b_mx = 0, b_i = 0, b_d = 0, b_s = 0;
    if (rs < tile_start) {
      const u64 g0 = rs / WGA_TILE;
      u64 p_mx = 0, p_i = 0, p_d = 0, p_s = 0;
      for (u64 k = g0 + tid; k < g; k += WGA_BLOCK) {
        const wga_tile_sum* t = a.tiles + k;
        const u64* v = (k == g0) ? t->tail : t->tot;
        p_mx += v[CLS_MX];
        p_i += v[CLS_I];
        p_d += v[CLS_D];
        p_s += v[CLS_S];
      }
      p_mx = wave_sum_u64(p_mx);
      p_i = wave_sum_u64(p_i);
      p_d = wave_sum_u64(p_d);
      p_s = wave_sum_u64(p_s);
      __syncthreads();
      if (lane == 0) {
        s_red[wave][0] = p_mx;
        s_red[wave][1] = p_i;
        s_red[wave][2] = p_d;
        s_red[wave][3] = p_s;
      }
      __syncthreads();
      for (int w2 = 0; w2 < 4; w2++) {
        b_mx += WGA_UNI64(s_red[w2][0]);
        b_i += WGA_UNI64(s_red[w2][1]);
        b_d += WGA_UNI64(s_red[w2][2]);
        b_s += WGA_UNI64(s_red[w2][3]);
      }
    }
    const u64 cb = b_mx + b_d;       /* target columns of this record before the segment */
    const u64 qb = b_mx + b_i + b_s; /* query bases consumed before it                   */

    const wga_class_sums cs = a.rec_sums[r];
    const u64 T_total = cs.mx + cs.d;          /* columns the CIGAR emits */
    const u64 Q_total = cs.mx + cs.i + cs.s;   /* query bases it consumes */
    RowSrc qs;
    qs.fa = a.q_fa;
    qs.fa_bytes = a.q_fa_bytes;
    qs.src_off = BASE ? a.q_src_off[r] : 0;
    qs.src_len = BASE ? a.q_src_len[r] : 0;
    qs.rc = a.strand_neg[r] != 0;
    /* edited length: String::drain / insert_str semantics (cigar.rs:769-786) */
    /* a record whose I / S ops take more than the slice holds (String::drain panics, reported below) has no row: the
     * difference must not wrap into a row of 2^64 bytes */
    const u64 row_len = BASE ? (qs.src_len + cs.d >= cs.i + cs.s ? qs.src_len + cs.d - (cs.i + cs.s) : 0ull) : T_total;
    const u64 skip = a.skip[r];
    u8* const dst = a.out + a.dst_off[r];
    u64* const bad_base = (u64*)&a.diag[r].bad_base_pos;
    u64* const panic_idx = (u64*)&a.diag[r].panic_op_idx;

    if (fast) {
      /* the same in every lane, but read from LDS: told to the compiler, or the whole row emitter below sits in
       * exec-masked control flow with its loop bounds in VGPRs */
      const u32 col_a = WGA_UNI32(s_col[ka]), seg_cols = WGA_UNI32(s_col[kb]) - col_a;
      const int ea = (int)WGA_UNI32(s_ev[ka]), eb = (int)WGA_UNI32(s_ev[kb]);
      const u32 adj_a = BASE ? WGA_UNI32(s_g_adj[ea]) : 0u;
      if (BASE) {
        /* drain(offset..offset+len) panics past the end of the string, insert_str(offset)
         * beyond it (cigar.rs:772,779): in slice terms, an I/S op needs q_before + len <= slice
         * length, a D op q_before <= slice length */
        const u32 d_a = WGA_UNI32(s_g_cum[ea]), is_a = d_a - adj_a;
        for (int e = 0; e < 4; e++) {
          u32 k = tid * 4u + (u32)e;
          if (k >= ka && k < kb) {
            /* the op is read again (it is not kept in registers across the row emitter), its own prefixes sit in
             * the event lists at its slot (s_ev = events before op k) */
            const u32 op = a.ops[tile_start + k];
            const u32 c = op_class(op & 15u);
            if (c == CLS_I || c == CLS_S || c == CLS_D) {
              const u32 ev = s_ev[k];
              const u32 e_d = s_g_cum[ev], e_is = e_d - s_g_adj[ev];
              u64 q_before = qb + (u64)(s_col[k] - col_a) - (u64)(e_d - d_a) + (u64)(e_is - is_a);
              u64 len = op >> 4;
              if (c != CLS_D && q_before + len > qs.src_len) atomicMin(panic_idx, tile_start + k - rs);
              if (c == CLS_D && q_before > qs.src_len) atomicMin(panic_idx, tile_start + k - rs);
            }
          }
        }
      }
      u64 x0 = cb > skip ? cb : skip;
      u64 x1 = cb + seg_cols < row_len ? cb + seg_cols : row_len;
      if (x1 > x0) {
        const u32 c_first = col_a + (u32)(x0 - cb);
        if (BASE) {
          RowDesc rd;
          rd.c_org = col_a;
          rd.G_col = s_g_col;
          rd.G_cum = s_g_cum;
          rd.G_adj = s_g_adj;
          rd.ga = ea;
          rd.gb = eb;
          rd.gcum_a = adj_a;
          rd.sbase = qb;
          rd.lowmask = s_lowmask;
          rd.tbl = s_tbl;
          rd.tsh = 0u;
          rd.gsh = gsh;
          rd.queue = s_queue;
          rowsrc_prepare(qs, qb);
          emit_row(dst + (x0 - skip), (u32)(x1 - x0), c_first, rd, qs, tid, WGA_BLOCK, bad_base);
        } else
          emit_symbols(dst + (x0 - skip), (u32)(x1 - x0), c_first, s_col, s_sym, (int)ka, (int)kb, s_tbl, gsh,
                       s_lowmask, tid, WGA_BLOCK);
      }
    }
    /* u64 fallback for tiles too wide for u32 columns: op-serial walk, every thread redundantly */
    if (!fast) {
      u64 x = cb, qp = qb;
      for (u64 k = cur; k < seg_end; k++) {
        const u32 op = a.ops[k];
        const u32 code = op & 15u;
        const u32 c = op_class(code);
        const u64 len = op >> 4;
        if (BASE && tid == 0) {
          if ((c == CLS_I || c == CLS_S) && qp + len > qs.src_len) atomicMin(panic_idx, k - rs);
          if (c == CLS_D && qp > qs.src_len) atomicMin(panic_idx, k - rs);
        }
        if (c == CLS_MX || c == CLS_D) {
          for (u64 j = tid; j < len; j += WGA_BLOCK) {
            u64 xx = x + j;
            if (xx >= skip && xx < row_len) {
              u8 v;
              if (BASE)
                v = (c == CLS_D) ? (u8)'-' : src_byte(qs, qp + j, bad_base);
              else
                v = (u8)(pseudo_symbol(code) & 0xFFu);
              dst[xx - skip] = v;
            }
          }
          x += len;
        }
        if (c == CLS_MX || c == CLS_I || c == CLS_S) qp += len;
      }
    }
    /* leftover query bases beyond the CIGAR stay at the end of the edited string */
    if (seg_end == re && BASE && row_len > T_total) {
      u64 x0 = T_total > skip ? T_total : skip;
      if (row_len > x0)
        emit_tail(dst + (x0 - skip), row_len - x0, Q_total + (x0 - T_total), qs, s_lowmask, s_queue, s_zero2, s_g_col, tid, WGA_BLOCK, bad_base);
    }
    cur = seg_end;
    r++;
  }
}
/* one block per tile of the batch: when the streaming row kernel is switched off ("pseudo_variant" 0) */
template <bool BASE>
__global__ __launch_bounds__(256, BASE ? WGA_K6_BLOCKS_BASE : WGA_K6_BLOCKS_SYM) void k_pafpseudo_fill(PseudoArgs a) {
  pseudo_tile<BASE>(a, xcd_tile_of_block());
}
/* the tiles the streaming row kernel (k_pafpseudo_stream / _sym, wga_kernels_k2s.h) leaves: giant tiles and, in base mode, records
 * whose slice is not exactly what their CIGAR consumes (leftover bases, drain / insert_str panics) and slices at a pool's edge */
template <bool BASE>
__global__ __launch_bounds__(256, BASE ? WGA_K6_BLOCKS_BASE : WGA_K6_BLOCKS_SYM) void k_pafpseudo_fill_list(PseudoArgs a) {
  const u32 n_list = *a.tile_count;
  for (u32 idx = blockIdx.x; idx < n_list; idx += gridDim.x) {
    pseudo_tile<BASE>(a, a.tile_list[idx]);
    __syncthreads(); /* the tile's LDS state is dead */
  }
}

/* ============================================================================================ */
/* K3 / K4: MAF column-pair walks                                                               */
/* ============================================================================================ */
/* One wave per record, 16 columns per lane and step (one byte-unaligned 16 B load per row, 1 KiB
 * per row and wave instruction).  Columns are classified four at a time on packed bytes: a byte
 * test leaves 0x80 in every byte that satisfies it, classes are small integers kept one per byte,
 * run starts are the bytes whose class differs from the byte before (the previous lane's last
 * class comes by DPP / shuffle, the previous step's by a carried value).  Counting is popcount;
 * only the few run starts are walked bit by bit.
 *   K3  cigar_cat_ext (cigar.rs:298-308): equal bytes -> '=' (also '-','-'; case-sensitive), else
 *       target gap -> I, else query gap -> D, else X.  Run entry = start_col << 3 | class
 *       (0 '=', 1 I, 2 D, 3 X).
 *   K4  cigar_cat_ext_caller (cigar.rs:314-328): gap tests first, so '-','-' is its own class W
 *       and splits runs.  Run entry = 3 u64: start_col << 3 | class (0 '=', 1 I, 2 D, 3 X, 4 W),
 *       non-gap target characters before the run, non-gap query characters before it. */
__device__ __forceinline__ u32 zero_bytes(u32 x) { /* 0x80 in every byte of x that is 0 (exact) */
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
__device__ __forceinline__ u32 popc32(u32 x) { return (u32)__builtin_popcount(x); }

#ifndef WGA_MAF_FOLD_STEPS
#define WGA_MAF_FOLD_STEPS 4095u /* the emulator build of the tests folds every few steps instead */
#endif
#ifndef WGA_K3_BLOCKS
#define WGA_K3_BLOCKS 6 /* blocks per CU the register budget of k_maf_pair_stat is sized for (two records per wave with prefetched rows: 8 spills six registers) */
#endif
struct MafWalkOut {
  u64 ncol[5], nrun[5]; /* columns / runs per class (wave totals, valid in every lane) */
  u64 runs;             /* runs in all */
  u64 t_nongap, q_nongap; /* caller walk: non-gap characters of the two rows (including the start values) */
};

/* The walk of columns [0, L) of the rows t, q.  For a PIECE of a longer row pair the caller passes the rows advanced
 * to the piece's first column, that column's index as col_bias (reported run starts are row-relative), the class of
 * the column in front of it (carry0; 0xFF at a row start) and the non-gap characters / runs of the row in front of
 * the piece (t_base0, q_base0 for the caller walk; rout already points at the piece's first run slot). */
/* ---- sixteen columns of a lane as one bit mask: bit 8e + d = column 4d + e (dword d, byte e) ---- */
__device__ __forceinline__ u32 maf_nonzero7(u32 x) { return ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x; } /* bit 7 of every byte that is not 0 (exact) */
__device__ __forceinline__ u32 maf_gather_bit7(const u32 y[4]) {
  u32 m = (y[0] >> 7) & 0x01010101u;
  m |= (y[1] >> 6) & 0x02020202u;
  m |= (y[2] >> 5) & 0x04040404u;
  m |= (y[3] >> 4) & 0x08080808u;
  return m;
}
/* the mask moved up by one column: column j takes column j - 1's bit, column 0 takes `first` (0 / 1) */
__device__ __forceinline__ u32 maf_prev_cols(u32 b, u32 first) { return (b << 8) | ((b >> 23) & 0xEu) | first; }
/* columns [0, nv) */
__device__ __forceinline__ u32 maf_valid_mask(u32 nv) {
  u32 v = 0u;
#pragma unroll
  for (u32 d = 0; d < 4u; d++) {
    const u32 n = nv > 4u * d ? (nv - 4u * d > 4u ? 4u : nv - 4u * d) : 0u; /* valid bytes of dword d */
    const u32 low = n >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n)) - 1u);
    v |= (0x01010101u << d) & low;
  }
  return v;
}

struct MafWalkStart {
  u64 col_bias, t_base, q_base;
  u32 carry;
};

/* this lane's 16 columns of the step that starts at column c0: two byte-unaligned 16-byte loads, or byte loads in the rows'
 * last, partial vector (never beyond the rows); zeros behind the rows' end */
struct MafStepRows {
  u32 t[4], q[4];
};
__device__ __forceinline__ void maf_load_step(const u8* __restrict__ t, const u8* __restrict__ q, u64 L, u64 c0, u32 lane,
                                              MafStepRows& r) {
  const u64 c = c0 + (u64)lane * 16u;
  const u32 nv = c >= L ? 0u : (L - c >= 16u ? 16u : (u32)(L - c));
#pragma unroll
  for (int d = 0; d < 4; d++) r.t[d] = r.q[d] = 0u;
  if (nv == 16u) {
    const u32x4_a1 a = *(const u32x4_a1*)(t + c), b = *(const u32x4_a1*)(q + c);
#pragma unroll
    for (int d = 0; d < 4; d++) r.t[d] = a[d], r.q[d] = b[d];
  } else if (nv) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const u32 j = 4u * (u32)d + (u32)e;
        if (j < nv) {
          r.t[d] |= (u32)t[c + j] << (8u * (u32)e);
          r.q[d] |= (u32)q[c + j] << (8u * (u32)e);
        }
      }
    }
  }
}

/* `first`: the rows of the first step, already loaded by the caller (the kernels fetch the next record's behind the work on the
 * current one); every further step's rows are fetched one step ahead. */
template <bool CALLER>
__device__ __forceinline__ void maf_walk(const u8* __restrict__ t, const u8* __restrict__ q, u64 L,
                                         u64* rout, MafWalkOut& out, const MafWalkStart st0, const bool have_first,
                                         const MafStepRows& first) {
  const u32 lane = threadIdx.x & 63u;
  constexpr int NC = CALLER ? 5 : 4;
  /* per-lane counters of classes 1..NC-1, columns in the low and run starts in the high 16 bits (a step adds at
   * most 16 to either): folded into wave-uniform totals before they can wrap.  Class 0 needs none: its columns
   * and runs are what is left of L and of the run total. */
  u32 pk[NC];
  u64 Ctot[NC], Rtot[NC];
#pragma unroll
  for (int k = 0; k < NC; k++) pk[k] = 0u, Ctot[k] = Rtot[k] = 0ull;
  u32 carry_cls = st0.carry; /* class of the column before this step's first one */
  u64 run_base = 0, t_base = st0.t_base, q_base = st0.q_base;
  u32 steps = 0;
  u32 acc_runs = 0, acc_t = 0, acc_q = 0; /* without a run list: this lane's run starts / non-gap characters since the last fold */
  MafStepRows nx = first;
  if (!have_first) maf_load_step(t, q, L, 0, lane, nx);
  u64 c0 = 0;
  /* one step; FULL: every lane holds 16 valid columns (all steps of a row pair but the last): the validity masks and the
   * search for the last valid column fold away */
  auto step = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    const u64 c = c0 + (u64)lane * 16u;
    const u32 nv = FULL ? 16u : (c >= L ? 0u : (L - c >= 16u ? 16u : (u32)(L - c))); /* valid columns of this lane */
    u32 tw[4], qw[4];
#pragma unroll
    for (int d = 0; d < 4; d++) tw[d] = nx.t[d], qw[d] = nx.q[d];
    if (c0 + 1024 < L) maf_load_step(t, q, L, c0 + 1024, lane, nx); /* wave-uniform: the next step's rows, behind this step's work */
    /* Sixteen columns as bit masks: bit 8e + d = column 4d + e (dword d, byte e).  A class is three bit planes (b0, b1, b2:
     * I = 001, D = 010, X = 011, W = 100, '=' = 000), every test and count below is one instruction for the lane's sixteen
     * columns instead of one per dword and class. */
    u32 yn[4], yt[4], yq[4];
    const u32 any_hi = ((tw[0] | tw[1] | tw[2]) | (tw[3] | qw[0] | qw[1]) | (qw[2] | qw[3])) & 0x80808080u;
    if (__ballot(any_hi != 0u) == 0ull) { /* wave-uniform; text: no byte has bit 7, so adding 0x7F per byte cannot carry into the next */
#pragma unroll
      for (int d = 0; d < 4; d++) {
        yn[d] = (tw[d] ^ qw[d]) + 0x7F7F7F7Fu;
        yt[d] = (tw[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
        yq[d] = (qw[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
      }
    } else {
#pragma unroll
      for (int d = 0; d < 4; d++) {
        yn[d] = maf_nonzero7(tw[d] ^ qw[d]);
        yt[d] = maf_nonzero7(tw[d] ^ 0x2D2D2D2Du);
        yq[d] = maf_nonzero7(qw[d] ^ 0x2D2D2D2Du);
      }
    }
    constexpr u32 ALL = 0x0F0F0F0Fu;
    const u32 V = FULL ? ALL : maf_valid_mask(nv);
    const u32 ne = maf_gather_bit7(yn), tng = maf_gather_bit7(yt) & V, qng = maf_gather_bit7(yq) & V; /* differ; t / q hold a base */
    const u32 tg = tng ^ V, qg = qng ^ V;
    u32 b0, b1, b2 = 0u;
    if (CALLER) { /* gap tests first (cigar.rs:314-328) */
      b0 = qng & (tg | (ne & tng));
      b1 = tng & (qg | (ne & qng));
      b2 = tg & qg;
    } else { /* equal bytes first (cigar.rs:298-308): two gaps are '=' */
      b0 = ne & (tg | qng) & V;
      b1 = ne & (qg | tng) & V;
    }
    const u32 cI = b0 & ~b1, cD = b1 & ~b0, cX = b0 & b1;
    pk[1] += popc32(cI);
    pk[2] += popc32(cD);
    pk[3] += popc32(cX);
    if (CALLER) pk[NC - 1] += popc32(b2);
    /* the class of the column in front of each column: within the lane a shift of the planes, the lane's first column takes
     * the last class of the lane below (of the step / piece in front for lane 0; 0xFF / 0xFE there match no class) */
    u32 my_last;
    if (FULL) {
      my_last = ((b0 >> 27) & 1u) | ((b1 >> 26) & 2u) | ((b2 >> 25) & 4u);
    } else {
      const u32 j = nv - 1u, pos = ((j & 3u) << 3) | (j >> 2);
      my_last = nv ? (((b0 >> pos) & 1u) | (((b1 >> pos) & 1u) << 1) | (((b2 >> pos) & 1u) << 2)) : 0xFEu;
    }
    u32 prev_last = __shfl_up(my_last, 1u);
    if (lane == 0) prev_last = carry_cls;
    u32 S = (b0 ^ maf_prev_cols(b0, prev_last & 1u)) | (b1 ^ maf_prev_cols(b1, (prev_last >> 1) & 1u));
    if (CALLER)
      S |= b2 ^ maf_prev_cols(b2, (prev_last >> 2) & 1u);
    else
      S |= (prev_last >> 2) & 1u; /* a row's / piece's first column after "no class" */
    S &= V;
    const u32 nst = popc32(S);
    pk[1] += popc32(S & cI) << 16;
    pk[2] += popc32(S & cD) << 16;
    pk[3] += popc32(S & cX) << 16;
    if (CALLER) pk[NC - 1] += popc32(S & b2) << 16;
    /* ordered run list: wave-exclusive offsets of the per-lane start counts (the totals alone when nothing is written) */
    u32 step_runs, t_excl = 0, q_excl = 0, t_tot = 0, q_tot = 0;
    if (rout) {
      const u32 incl = wave_incl_scan_u32(nst);
      step_runs = wave_last_u32(incl);
      if (CALLER) {
        const u32 tnc = popc32(tng), qnc = popc32(qng);
        const u32 ti = wave_incl_scan_u32(tnc), qi = wave_incl_scan_u32(qnc);
        t_excl = ti - tnc;
        q_excl = qi - qnc;
        t_tot = wave_last_u32(ti);
        q_tot = wave_last_u32(qi);
      }
      if (nst) {
        u64 slot = run_base + (u64)(incl - nst);
        u32 tb = 0, qb = 0; /* non-gap bytes of this lane before the dword being walked */
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 m = (S >> d) & 0x01010101u; /* dword d's columns, byte e at bit 8e */
          const u32 td = (tng >> d) & 0x01010101u, qd = (qng >> d) & 0x01010101u;
          while (m) {
            const u32 bit = (u32)__builtin_ctz(m); /* 0, 8, 16 or 24 */
            const u32 pos = bit + (u32)d;
            const u32 k = ((b0 >> pos) & 1u) | (((b1 >> pos) & 1u) << 1) | (((b2 >> pos) & 1u) << 2);
            const u64 col = st0.col_bias + c + 4u * (u32)d + (bit >> 3);
            if (CALLER) {
              const u32 bm = (1u << bit) - 1u; /* bytes below */
              u64* e = rout + 3 * slot;
              e[0] = (col << 3) | (u64)k;
              e[1] = t_base + t_excl + tb + popc32(td & bm);
              e[2] = q_base + q_excl + qb + popc32(qd & bm);
            } else {
              rout[slot] = (col << 3) | (u64)k;
            }
            slot++;
            m &= m - 1u;
          }
          tb += popc32(td);
          qb += popc32(qd);
        }
      }
    } else { /* totals only: the lanes keep their own sums, added up when the walk ends (or before they could wrap) */
      step_runs = 0u;
      acc_runs += nst;
      if (CALLER) {
        acc_t += popc32(tng);
        acc_q += popc32(qng);
      }
    }
    run_base += (u64)step_runs;
    t_base += t_tot;
    q_base += q_tot;
    /* the last valid column of this step is in the last lane that has any */
    if (FULL) {
      carry_cls = wave_last_u32(my_last);
    } else {
      const u64 has = __ballot(nv != 0u);
      const int last_lane = 63 - (int)__builtin_clzll(has); /* has != 0 inside the loop */
      carry_cls = __shfl(my_last, last_lane);
    }
    if (++steps == WGA_MAF_FOLD_STEPS) { /* 16 x 4095 < 2^16: fold the lane counters before a half can wrap */
#pragma unroll
      for (int k = 1; k < NC; k++) {
        Ctot[k] += wave_sum_u32(pk[k] & 0xFFFFu);
        Rtot[k] += wave_sum_u32(pk[k] >> 16);
        pk[k] = 0u;
      }
      run_base += (u64)wave_sum_u32(acc_runs);
      t_base += (u64)wave_sum_u32(acc_t);
      q_base += (u64)wave_sum_u32(acc_q);
      acc_runs = acc_t = acc_q = 0u;
      steps = 0;
    }
  };
  for (; c0 < L; c0 += 1024) {
    if (c0 + 1024 <= L) /* wave-uniform */
      step(std::true_type{});
    else
      step(std::false_type{});
  }
  if (!rout) { /* wave-uniform */
    run_base += (u64)wave_sum_u32(acc_runs);
    if (CALLER) {
      t_base += (u64)wave_sum_u32(acc_t);
      q_base += (u64)wave_sum_u32(acc_q);
    }
  }
  /* class 0 columns / runs = all minus the others */
  u64 C[NC], R[NC];
  C[0] = L;
  R[0] = run_base;
  if (L < 65536u && L <= (u64)(WGA_MAF_FOLD_STEPS - 1u) * 1024u) { /* no fold happened and every wave total fits 16 bits: both halves in one scan */
#pragma unroll
    for (int k = 1; k < NC; k++) {
      const u32 a = wave_sum_u32(pk[k]);
      C[k] = a & 0xFFFFu, R[k] = a >> 16;
    }
  } else {
#pragma unroll
    for (int k = 1; k < NC; k++) {
      C[k] = Ctot[k] + wave_sum_u32(pk[k] & 0xFFFFu);
      R[k] = Rtot[k] + wave_sum_u32(pk[k] >> 16);
    }
  }
  u64 oc = 0, orn = 0;
#pragma unroll
  for (int k = 1; k < NC; k++) oc += C[k], orn += R[k];
  out.ncol[0] = C[0] - oc;
  out.nrun[0] = R[0] - orn;
#pragma unroll
  for (int k = 1; k < NC; k++) out.ncol[k] = C[k], out.nrun[k] = R[k];
  if (!CALLER) out.ncol[4] = out.nrun[4] = 0;
  out.runs = run_base;
  out.t_nongap = t_base;
  out.q_nongap = q_base;
}

#define WGA_MAF_PAIR_MAX 65000ull /* blocks the pair walk takes: every total fits sixteen bits */
/* ... and only where one stream is fewer steps than two walks (its lanes carry a block id, two sets of totals) */
__device__ __forceinline__ bool maf_pair_pays(u64 L0, u64 L1) {
  if (L0 >= WGA_MAF_PAIR_MAX || L1 >= WGA_MAF_PAIR_MAX) return false;
  const u64 P = (L0 + 15ull) & ~15ull;
  return (L0 + 1023ull) / 1024ull + (L1 + 1023ull) / 1024ull > (P + L1 + 1023ull) / 1024ull;
}
/* ---- two short blocks as ONE column stream -----------------------------------------------------------------------
 * A block of 1 500 columns is one full step and one of 476 columns — and a step costs its instructions whatever the number of
 * lanes that hold columns.  Two blocks of a wave are therefore walked as one stream: block A's columns, padded to a multiple of
 * sixteen (so that a lane's sixteen columns belong to ONE block), then block B's; two blocks of 1 500 columns are three steps
 * instead of four.  A lane knows its block (`rb`), its first column in it and how many of its columns are valid; the class of
 * the column in front of B's first lane is "none"; every lane keeps two sets of totals, A's and B's (a lane serves A in one step
 * and B in another); with run lists, the lanes' slots and non-gap prefixes start again at B's first lane.  The step itself — masks,
 * planes, starts, counts — is maf_walk's.  Both blocks are at most `long_cols` columns (u32 arithmetic, no folds: < 2^16 per total). */
__device__ __forceinline__ void maf_load_lane(const u8* __restrict__ tp, const u8* __restrict__ qp, u32 nv, MafStepRows& r) {
#pragma unroll
  for (int d = 0; d < 4; d++) r.t[d] = r.q[d] = 0u;
  if (nv == 16u) {
    const u32x4_a1 a = *(const u32x4_a1*)tp, b = *(const u32x4_a1*)qp;
#pragma unroll
    for (int d = 0; d < 4; d++) r.t[d] = a[d], r.q[d] = b[d];
  } else if (nv) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const u32 j = 4u * (u32)d + (u32)e;
        if (j < nv) {
          r.t[d] |= (u32)tp[j] << (8u * (u32)e);
          r.q[d] |= (u32)qp[j] << (8u * (u32)e);
        }
      }
    }
  }
}
template <bool CALLER>
__device__ __forceinline__ void maf_walk_pair(const u8* __restrict__ tA, const u8* __restrict__ qA, const u32 LA, u64* const routA,
                                              const u8* __restrict__ tB, const u8* __restrict__ qB, const u32 LB, u64* const routB,
                                              MafWalkOut& outA, MafWalkOut& outB) {
  const u32 lane = threadIdx.x & 63u;
  constexpr int NC = CALLER ? 5 : 4;
  const bool lists = routA != nullptr; /* both or neither (wave-uniform) */
  const u32 P = (LA + 15u) & ~15u, G = P + LB; /* B's first column in the stream, the stream's length */
  u32 pkA[NC], pkB[NC];                        /* per class: columns in the low, run starts in the high 16 bits */
#pragma unroll
  for (int k = 0; k < NC; k++) pkA[k] = pkB[k] = 0u;
  u32 accA = 0, accB = 0, acctA = 0, acctB = 0, accqA = 0, accqB = 0; /* without lists: run starts / non-gap characters of the lane */
  u32 runA = 0, runB = 0, tbA = 0, tbB = 0, qbA = 0, qbB = 0;         /* with lists: runs / non-gap characters in front of the step */
  u32 carry_cls = 0xFFu;
  auto lane_of = [&](u32 g0, bool& rb, u32& crel, u32& nv) { /* this lane's sixteen columns of the step at stream column g0 */
    const u32 gl = g0 + 16u * lane;
    rb = gl >= P;
    crel = rb ? gl - P : gl;
    const u32 Lr = rb ? LB : LA;
    nv = crel >= Lr ? 0u : (Lr - crel >= 16u ? 16u : Lr - crel);
  };
  MafStepRows nx;
  {
    bool rb;
    u32 crel, nv;
    lane_of(0u, rb, crel, nv);
    maf_load_lane((rb ? tB : tA) + crel, (rb ? qB : qA) + crel, nv, nx);
  }
  for (u32 g0 = 0; g0 < G; g0 += 1024u) {
    bool rb;
    u32 crel, nv;
    lane_of(g0, rb, crel, nv);
    u32 tw[4], qw[4];
#pragma unroll
    for (int d = 0; d < 4; d++) tw[d] = nx.t[d], qw[d] = nx.q[d];
    if (g0 + 1024u < G) { /* wave-uniform: the next step's rows, behind this step's work */
      bool rb2;
      u32 crel2, nv2;
      lane_of(g0 + 1024u, rb2, crel2, nv2);
      maf_load_lane((rb2 ? tB : tA) + crel2, (rb2 ? qB : qA) + crel2, nv2, nx);
    }
    u32 yn[4], yt[4], yq[4];
    const u32 any_hi = ((tw[0] | tw[1] | tw[2]) | (tw[3] | qw[0] | qw[1]) | (qw[2] | qw[3])) & 0x80808080u;
    if (__ballot(any_hi != 0u) == 0ull) {
#pragma unroll
      for (int d = 0; d < 4; d++) {
        yn[d] = (tw[d] ^ qw[d]) + 0x7F7F7F7Fu;
        yt[d] = (tw[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
        yq[d] = (qw[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
      }
    } else {
#pragma unroll
      for (int d = 0; d < 4; d++) {
        yn[d] = maf_nonzero7(tw[d] ^ qw[d]);
        yt[d] = maf_nonzero7(tw[d] ^ 0x2D2D2D2Du);
        yq[d] = maf_nonzero7(qw[d] ^ 0x2D2D2D2Du);
      }
    }
    const u32 V = maf_valid_mask(nv);
    const u32 ne = maf_gather_bit7(yn), tng = maf_gather_bit7(yt) & V, qng = maf_gather_bit7(yq) & V;
    const u32 tg = tng ^ V, qg = qng ^ V;
    u32 b0, b1, b2 = 0u;
    if (CALLER) {
      b0 = qng & (tg | (ne & tng));
      b1 = tng & (qg | (ne & qng));
      b2 = tg & qg;
    } else {
      b0 = ne & (tg | qng) & V;
      b1 = ne & (qg | tng) & V;
    }
    const u32 cI = b0 & ~b1, cD = b1 & ~b0, cX = b0 & b1;
    u32 my_last;
    {
      const u32 j = nv - 1u, pos = ((j & 3u) << 3) | (j >> 2);
      my_last = nv ? (((b0 >> pos) & 1u) | (((b1 >> pos) & 1u) << 1) | (((b2 >> pos) & 1u) << 2)) : 0xFEu;
    }
    u32 prev_last = __shfl_up(my_last, 1u);
    if (lane == 0) prev_last = carry_cls;
    if (crel == 0u) prev_last = 0xFFu; /* a block's first column */
    u32 S = (b0 ^ maf_prev_cols(b0, prev_last & 1u)) | (b1 ^ maf_prev_cols(b1, (prev_last >> 1) & 1u));
    if (CALLER)
      S |= b2 ^ maf_prev_cols(b2, (prev_last >> 2) & 1u);
    else
      S |= (prev_last >> 2) & 1u;
    S &= V;
    const u32 nst = popc32(S);
    u32 x[NC];
    x[0] = 0u;
    x[1] = popc32(cI) | (popc32(S & cI) << 16);
    x[2] = popc32(cD) | (popc32(S & cD) << 16);
    x[3] = popc32(cX) | (popc32(S & cX) << 16);
    if (CALLER) x[NC - 1] = popc32(b2) | (popc32(S & b2) << 16);
#pragma unroll
    for (int k = 1; k < NC; k++) {
      pkA[k] += rb ? 0u : x[k];
      pkB[k] += rb ? x[k] : 0u;
    }
    const u32 tnc = CALLER ? popc32(tng) : 0u, qnc = CALLER ? popc32(qng) : 0u;
    if (lists) { /* wave-uniform */
      /* the lanes of A in this step come first; B's slots and prefixes start again behind them */
      const u32 nA = g0 >= P ? 0u : ((P - g0) >> 4 > 64u ? 64u : (P - g0) >> 4);
      const u32 incl = wave_incl_scan_u32(nst);
      const u32 runs_all = wave_last_u32(incl), runs_A = nA ? wave_get_u32_dyn(incl, nA - 1u) : 0u;
      u32 t_excl = 0, q_excl = 0, t_all = 0, q_all = 0, t_A = 0, q_A = 0;
      if (CALLER) {
        const u32 ti = wave_incl_scan_u32(tnc), qi = wave_incl_scan_u32(qnc);
        t_excl = ti - tnc;
        q_excl = qi - qnc;
        t_all = wave_last_u32(ti);
        q_all = wave_last_u32(qi);
        t_A = nA ? wave_get_u32_dyn(ti, nA - 1u) : 0u;
        q_A = nA ? wave_get_u32_dyn(qi, nA - 1u) : 0u;
      }
      if (nst) {
        u64* const rout = rb ? routB : routA;
        u32 slot = (rb ? runB - runs_A : runA) + (incl - nst);
        u32 tb = (rb ? tbB - t_A : tbA) + t_excl, qb = (rb ? qbB - q_A : qbA) + q_excl;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 m = (S >> d) & 0x01010101u;
          const u32 td = (tng >> d) & 0x01010101u, qd = (qng >> d) & 0x01010101u;
          while (m) {
            const u32 bit = (u32)__builtin_ctz(m);
            const u32 pos = bit + (u32)d;
            const u32 k = ((b0 >> pos) & 1u) | (((b1 >> pos) & 1u) << 1) | (((b2 >> pos) & 1u) << 2);
            const u64 col = (u64)(crel + 4u * (u32)d + (bit >> 3));
            if (CALLER) {
              const u32 bm = (1u << bit) - 1u;
              u64* e = rout + 3 * (u64)slot;
              e[0] = (col << 3) | (u64)k;
              e[1] = (u64)(tb + popc32(td & bm));
              e[2] = (u64)(qb + popc32(qd & bm));
            } else {
              rout[slot] = (col << 3) | (u64)k;
            }
            slot++;
            m &= m - 1u;
          }
          tb += popc32(td);
          qb += popc32(qd);
        }
      }
      runA += runs_A;
      runB += runs_all - runs_A;
      tbA += t_A;
      tbB += t_all - t_A;
      qbA += q_A;
      qbB += q_all - q_A;
    } else {
      accA += rb ? 0u : nst;
      accB += rb ? nst : 0u;
      if (CALLER) {
        acctA += rb ? 0u : tnc;
        acctB += rb ? tnc : 0u;
        accqA += rb ? 0u : qnc;
        accqB += rb ? qnc : 0u;
      }
    }
    { /* the last valid column of this step is in the last lane that has any */
      const u64 has = __ballot(nv != 0u);
      if (has) carry_cls = __shfl(my_last, 63 - (int)__builtin_clzll(has));
    }
  }
  if (!lists) {
    runA = wave_sum_u32(accA);
    runB = wave_sum_u32(accB);
    if (CALLER) {
      tbA = wave_sum_u32(acctA);
      tbB = wave_sum_u32(acctB);
      qbA = wave_sum_u32(accqA);
      qbB = wave_sum_u32(accqB);
    }
  }
  u64 ocA = 0, orA = 0, ocB = 0, orB = 0;
#pragma unroll
  for (int k = 1; k < NC; k++) {
    const u32 a = wave_sum_u32(pkA[k]), b = wave_sum_u32(pkB[k]); /* < 2^16 in either half: the blocks are short */
    outA.ncol[k] = a & 0xFFFFu, outA.nrun[k] = a >> 16;
    outB.ncol[k] = b & 0xFFFFu, outB.nrun[k] = b >> 16;
    ocA += outA.ncol[k], orA += outA.nrun[k], ocB += outB.ncol[k], orB += outB.nrun[k];
  }
  outA.ncol[0] = (u64)LA - ocA, outA.nrun[0] = (u64)runA - orA;
  outB.ncol[0] = (u64)LB - ocB, outB.nrun[0] = (u64)runB - orB;
  if (!CALLER) outA.ncol[4] = outA.nrun[4] = outB.ncol[4] = outB.nrun[4] = 0;
  outA.runs = runA, outB.runs = runB;
  outA.t_nongap = tbA, outA.q_nongap = qbA, outB.t_nongap = tbB, outB.q_nongap = qbB;
}

__device__ __forceinline__ void maf_pair_store(const MafWalkOut& w, bool neg, wga_cigar_counts* cnt, u64* run_cnt, u32 lane);
__device__ __forceinline__ void maf_pair_one(const u8* __restrict__ t, const u8* __restrict__ q, u64 L, u64* rout, bool neg,
                                             const MafStepRows& first, wga_cigar_counts* cnt, u64* run_cnt, u32 lane) {
  MafWalkOut w;
  maf_walk<false>(t, q, L, rout, w, MafWalkStart{0, 0, 0, 0xFFu}, true, first);
  maf_pair_store(w, neg, cnt, run_cnt, lane);
}
__device__ __forceinline__ void maf_pair_store(const MafWalkOut& w, bool neg, wga_cigar_counts* cnt, u64* run_cnt, u32 lane) {
  /* the 11 counters leave from lanes 0..10, one field per lane (as in K1): one 88-byte store per record */
  const u64 z = 0;
  u64 v = 0;
  v = lane_put_u64<0u>(v, w.ncol[0], lane);
  v = lane_put_u64<1u>(v, w.ncol[3], lane);
  v = lane_put_u64<2u>(v, neg ? z : w.nrun[1], lane);
  v = lane_put_u64<3u>(v, neg ? z : w.ncol[1], lane);
  v = lane_put_u64<4u>(v, neg ? z : w.nrun[2], lane);
  v = lane_put_u64<5u>(v, neg ? z : w.ncol[2], lane);
  v = lane_put_u64<6u>(v, neg ? w.nrun[1] : z, lane);
  v = lane_put_u64<7u>(v, neg ? w.ncol[1] : z, lane);
  v = lane_put_u64<8u>(v, neg ? w.nrun[2] : z, lane);
  v = lane_put_u64<9u>(v, neg ? w.ncol[2] : z, lane);
  v = lane_put_u64<10u>(v, neg ? (u64)1 : z, lane);
  if (lane < 11u) ((u64*)cnt)[lane] = v;
  if (lane == 0 && run_cnt) *run_cnt = w.runs;
}

/* Two consecutive records per wave: the offsets of both are fetched together and the second record's first rows travel while
 * the first record is walked — three dependent round trips (offsets, rows, every further step) stood in front of the work
 * of a 1 500-column block, the step loop above and this pairing leave one. */
__global__ __launch_bounds__(256, WGA_K3_BLOCKS) void k_maf_pair_stat(u32 n, const u8* __restrict__ rows,
                                                       const u64* t_off, const u64* q_off,
                                                       const u64* cols, const u8* strand_neg,
                                                       wga_cigar_counts* counts, u64* run_cnt,
                                                       u64* runs, const u64* run_off, u64 long_cols) {
  const u32 lane = threadIdx.x & 63u;
  const u64 i0 = ((u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x)) * 2u;
  if (i0 >= n) return;
  const bool two = i0 + 1u < n;
  const u64 i1 = two ? i0 + 1u : i0;
  const u64 L0 = cols[i0], L1 = cols[i1];
  const u8 *t0 = rows + t_off[i0], *q0 = rows + q_off[i0], *t1 = rows + t_off[i1], *q1 = rows + q_off[i1];
  const bool neg0 = strand_neg[i0] != 0, neg1 = strand_neg[i1] != 0;
  u64 *r0 = (u64*)0, *r1 = (u64*)0;
  if (runs) r0 = runs + run_off[i0], r1 = runs + run_off[i1];
  const bool do0 = L0 <= long_cols, do1 = two && L1 <= long_cols; /* a long block: walked piece by piece (k_maf_piece_walk) */
  if (do0 && do1 && maf_pair_pays(L0, L1)) { /* wave-uniform: two short blocks as one column stream */
    MafWalkOut wA, wB;
    maf_walk_pair<false>(t0, q0, (u32)L0, r0, t1, q1, (u32)L1, r1, wA, wB);
    maf_pair_store(wA, neg0, counts + i0, run_cnt ? run_cnt + i0 : (u64*)0, lane);
    maf_pair_store(wB, neg1, counts + i1, run_cnt ? run_cnt + i1 : (u64*)0, lane);
    return;
  }
  MafStepRows f0, f1;
  maf_load_step(t0, q0, do0 ? L0 : 0, 0, lane, f0);
  maf_load_step(t1, q1, do1 ? L1 : 0, 0, lane, f1);
  if (do0) maf_pair_one(t0, q0, L0, r0, neg0, f0, counts + i0, run_cnt ? run_cnt + i0 : (u64*)0, lane); /* wave-uniform */
  if (do1) maf_pair_one(t1, q1, L1, r1, neg1, f1, counts + i1, run_cnt ? run_cnt + i1 : (u64*)0, lane);
}

__global__ __launch_bounds__(256) void k_maf_call_runs(u32 n, const u8* __restrict__ rows,
                                                       const u64* t_off, const u64* q_off,
                                                       const u64* cols, u64* run_cnt, u64* runs,
                                                       const u64* run_off, u64 long_cols) {
  const u32 lane = threadIdx.x & 63u;
  const u64 i0 = ((u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x)) * 2u;
  if (i0 >= n) return;
  const bool two = i0 + 1u < n;
  const u64 i1 = two ? i0 + 1u : i0;
  const u64 L0 = cols[i0], L1 = cols[i1];
  const u8 *t0 = rows + t_off[i0], *q0 = rows + q_off[i0], *t1 = rows + t_off[i1], *q1 = rows + q_off[i1];
  u64 *r0 = (u64*)0, *r1 = (u64*)0;
  if (runs) r0 = runs + 3 * run_off[i0], r1 = runs + 3 * run_off[i1];
  const bool do0 = L0 <= long_cols, do1 = two && L1 <= long_cols;
  if (do0 && do1 && maf_pair_pays(L0, L1)) { /* wave-uniform: two short blocks as one column stream */
    MafWalkOut wA, wB;
    maf_walk_pair<true>(t0, q0, (u32)L0, r0, t1, q1, (u32)L1, r1, wA, wB);
    if (lane == 0 && run_cnt) run_cnt[i0] = wA.runs, run_cnt[i1] = wB.runs;
    return;
  }
  MafStepRows f0, f1;
  maf_load_step(t0, q0, do0 ? L0 : 0, 0, lane, f0);
  maf_load_step(t1, q1, do1 ? L1 : 0, 0, lane, f1);
  if (do0) { /* wave-uniform */
    MafWalkOut w;
    maf_walk<true>(t0, q0, L0, r0, w, MafWalkStart{0, 0, 0, 0xFFu}, true, f0);
    if (lane == 0 && run_cnt) run_cnt[i0] = w.runs;
  }
  if (do1) {
    MafWalkOut w;
    maf_walk<true>(t1, q1, L1, r1, w, MafWalkStart{0, 0, 0, 0xFFu}, true, f1);
    if (lane == 0 && run_cnt) run_cnt[i1] = w.runs;
  }
}

template <bool CALLER>
__device__ __forceinline__ void maf_walk(const u8* __restrict__ t, const u8* __restrict__ q, u64 L, u64* rout, MafWalkOut& out,
                                         const MafWalkStart st0 = MafWalkStart{0, 0, 0, 0xFFu}) {
  MafStepRows none;
#pragma unroll
  for (int d = 0; d < 4; d++) none.t[d] = none.q[d] = 0u;
  maf_walk<CALLER>(t, q, L, rout, out, st0, false, none);
}

/* ---- long blocks: the same walks, piece by piece ---------------------------------------------------------------
 * A block of 10^8 columns (`call --chunk-size` exists because such blocks do) is no work for
 * one wave.  Nothing in the walk is sequential: the class of a column is a function of that column, a run starts
 * where the class differs from the column before, the counters are sums and the caller walk's "non-gap characters
 * before the run" are prefix sums.  A block beyond `long_cols` columns is cut into pieces of `piece_cols`; every
 * piece is one wave's walk (k_maf_piece_walk, a persistent grid over the piece list), started with the class of the
 * column in front of it; a first pass leaves every piece's run and non-gap totals, an exclusive scan turns them into
 * the piece's first run slot and start values, and the fill pass writes the runs in order.  The counters of a long
 * block are added up with one atomic per field and piece. */
struct wga_maf_piece_tot {
  u64 runs, t_nongap, q_nongap;
};
/* pieces per record (0 for the records the one-wave kernels keep); long records get their counters zeroed */
__global__ __launch_bounds__(256) void k_maf_piece_counts(u32 n, const u64* cols, u64 long_cols, u64 piece_cols,
                                                          u64* npieces, wga_cigar_counts* counts, u64* run_cnt) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 c = cols[i];
  const bool lng = c > long_cols;
  npieces[i] = lng ? (c + piece_cols - 1) / piece_cols : 0;
  if (lng) {
    if (counts) {
      u64* f = (u64*)(counts + i);
      for (int k = 0; k < 11; k++) f[k] = 0;
    }
    if (run_cnt) run_cnt[i] = 0;
  }
}

__device__ __forceinline__ u32 maf_col_class(u8 tc, u8 qc, bool caller) {
  const bool tg = tc == (u8)'-', qg = qc == (u8)'-';
  if (caller) return (tg && qg) ? 4u : tg ? 1u : qg ? 2u : (tc == qc ? 0u : 3u);
  return tc == qc ? 0u : tg ? 1u : qg ? 2u : 3u;
}

struct ScanPieceTot { /* three exclusive scans in one pass over the piece totals */
  const wga_maf_piece_tot* in;
  int field;
  __device__ u64 operator()(u32 p) const { return field == 0 ? in[p].runs : field == 1 ? in[p].t_nongap : in[p].q_nongap; }
};

/* MODE 0: count (piece totals; K3 also adds the piece's counters to its record; run_cnt[i] += runs).
 * MODE 1: fill (runs written at the piece's slot).  piece_off = exclusive scan of npieces (n + 1 entries);
 * ex_runs / ex_t / ex_q = exclusive scans of the piece totals (fill only). */
template <bool CALLER, int MODE>
__global__ __launch_bounds__(256) void k_maf_piece_walk(u32 n, const u8* __restrict__ rows, const u64* t_off,
                                                        const u64* q_off, const u64* cols, const u8* strand_neg,
                                                        const u64* piece_off, u64 piece_cols,
                                                        wga_maf_piece_tot* ptot, const u64* ex_runs, const u64* ex_t,
                                                        const u64* ex_q, wga_cigar_counts* counts, u64* run_cnt,
                                                        u64* runs, const u64* run_off) {
  const u32 lane = threadIdx.x & 63u;
  const u64 n_pieces = piece_off[n];
  const u64 n_waves = (u64)gridDim.x * 4u;
  for (u64 p = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x); p < n_pieces; p += n_waves) {
    /* record of piece p: last i with piece_off[i] <= p (wave-uniform bisection) */
    u32 lo = 0, hi = n;
    while (hi - lo > 1u) {
      const u32 mid = lo + ((hi - lo) >> 1);
      if (piece_off[mid] <= p)
        lo = mid;
      else
        hi = mid;
    }
    const u32 i = lo;
    const u64 p0 = piece_off[i];
    const u64 c0 = (p - p0) * piece_cols;
    const u64 L = cols[i] - c0 < piece_cols ? cols[i] - c0 : piece_cols;
    const u8* t = rows + t_off[i];
    const u8* q = rows + q_off[i];
    MafWalkStart st;
    st.col_bias = c0;
    st.carry = c0 ? maf_col_class(t[c0 - 1], q[c0 - 1], CALLER) : 0xFFu;
    st.t_base = st.q_base = 0;
    u64* rout = (u64*)0;
    if (MODE == 1) {
      const u64 first = ex_runs[p] - ex_runs[p0]; /* runs of this record in front of the piece */
      rout = runs + (CALLER ? 3u : 1u) * (run_off[i] + first);
      if (CALLER) {
        st.t_base = ex_t[p] - ex_t[p0];
        st.q_base = ex_q[p] - ex_q[p0];
      }
    }
    MafWalkOut w;
    maf_walk<CALLER>(t + c0, q + c0, L, rout, w, st);
    if (MODE == 0) {
      if (lane == 0) {
        wga_maf_piece_tot pt;
        pt.runs = w.runs;
        pt.t_nongap = w.t_nongap;
        pt.q_nongap = w.q_nongap;
        ptot[p] = pt;
        if (run_cnt) atomicAdd(run_cnt + i, w.runs);
      }
      if (!CALLER && counts) {
        const bool neg = strand_neg[i] != 0;
        const u64 z = 0;
        u64 v = 0;
        v = lane_put_u64<0u>(v, w.ncol[0], lane);
        v = lane_put_u64<1u>(v, w.ncol[3], lane);
        v = lane_put_u64<2u>(v, neg ? z : w.nrun[1], lane);
        v = lane_put_u64<3u>(v, neg ? z : w.ncol[1], lane);
        v = lane_put_u64<4u>(v, neg ? z : w.nrun[2], lane);
        v = lane_put_u64<5u>(v, neg ? z : w.ncol[2], lane);
        v = lane_put_u64<6u>(v, neg ? w.nrun[1] : z, lane);
        v = lane_put_u64<7u>(v, neg ? w.ncol[1] : z, lane);
        v = lane_put_u64<8u>(v, neg ? w.nrun[2] : z, lane);
        v = lane_put_u64<9u>(v, neg ? w.ncol[2] : z, lane);
        v = lane_put_u64<10u>(v, (neg && p == p0) ? (u64)1 : z, lane); /* inv_event = 1 per '-' record: its first piece */
        if (lane < 11u && v) atomicAdd((u64*)(counts + i) + lane, v);
      }
    }
  }
}

/* ============================================================================================ */
/* K7: PAF call op walk                                                                         */
/* ============================================================================================ */
/* call_within_var_paf (caller.rs:610-822) walks the op stream with two running positions and an
 * `after_m` flag and raises events at X ops (when SNPs are asked for) and at I / D ops longer
 * than the cutoff that directly follow an M / = / X op.  Per record one wave scans the ops 64 at
 * a time: exclusive u64 prefix sums of the target / query advance, `after_m` from the previous
 * op's code, compaction of the event ops by ballot.  The walk stops at the first op that is not
 * M = X I D (the reference's fold keeps its Err and skips the rest, :673,815-819).
 * A length >= 2^28 is packed as a head op plus continuation pieces (codes 9 / 10): the head is
 * flagged when `len > svlen` or a continuation follows; the host applies the cutoff to the sum.
 * Event entry = 3 u64: op index in the record, target advance before it, query advance before it. */
__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v, u32 lane) {
#pragma unroll
  for (u32 d = 1; d < 64; d <<= 1) {
    const u64 o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

/* The walk of ops [a, b) of one record by one wave, 4 consecutive ops per lane and 256 per step (a is a multiple of 256): running
 * target / query positions and event count start from `st`, the op in front of a gives `after_m`, the op behind a step
 * tells whether an indel goes on in a continuation piece.  Events go to eout + 3 * (running count) when eout is given.
 * Returns the sums over the walked ops in *tot (events counted up to the walk's stop) and the record-relative index of
 * the first op outside M = X I D in *bad_at (WGA_NONE: none) — the walk stops there, as the reference's fold does before
 * it discards the error (caller.rs:673,815-819). */
struct PafCallState {
  u64 t, q, e;
};
/* POS = false: only the sums are wanted (a count pass): no prefix scans, every lane keeps its own sums and the wave adds them
 * up once when the walk ends */
template <bool POS>
__device__ __forceinline__ void paf_call_walk(const u32* __restrict__ rec, u64 nops, u64 a, u64 b, u64 svlen, u32 snp,
                                              PafCallState st, u64* eout, u32 lane, PafCallState* tot, u64* bad_at) {
  u64 t_base = st.t, q_base = st.q, e_base = st.e;
  u64 acc_t = 0, acc_q = 0; /* !POS: this lane's target / query advance ... */
  u32 acc_e = 0;            /* ... and events */
  u32 carry_code = a ? (rec[a - 1] & 15u) : 0xFu;
  *bad_at = WGA_NONE;
  u32 wnext[4]; /* the next step's ops travel behind the work on this step's */
#pragma unroll
  for (int e = 0; e < 4; e++) wnext[e] = a + (u64)lane * 4u + (u64)e < b ? rec[a + (u64)lane * 4u + e] : 0xFu; /* 0xF: no op */
  for (u64 k0 = a; k0 < b; k0 += 256) {
    const u64 kb = k0 + (u64)lane * 4u;
    u32 w[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      w[e] = wnext[e];
      wnext[e] = kb + 256u + (u64)e < b ? rec[kb + 256u + e] : 0xFu;
    }
    u32 code[4], len[4];
    bool valid[4], isi[4], isd[4];
    u32 firstbad = 4u; /* first op of this lane outside M = X I D (and inside the range) */
#pragma unroll
    for (int e = 3; e >= 0; e--) {
      code[e] = w[e] & 15u;
      len[e] = w[e] >> 4;
      const bool mlike = code[e] == WGA_OP_M || code[e] == WGA_OP_EQ || code[e] == WGA_OP_X;
      isi[e] = code[e] == WGA_OP_I || code[e] == WGA_OP_I_CONT;
      isd[e] = code[e] == WGA_OP_D || code[e] == WGA_OP_D_CONT;
      valid[e] = mlike || isi[e] || isd[e];
      if (kb + (u64)e < b && !valid[e]) firstbad = (u32)e;
    }
    /* the walk stops at the first bad op of the record: ops at or after it are dead */
    const u64 badm = __ballot(firstbad < 4u);
    u32 stop = 0xFFFFFFFFu; /* index inside this step */
    if (badm) {
      const int bl = (int)__builtin_ctzll(badm);
      stop = (u32)bl * 4u + (u32)__shfl((int)firstbad, bl);
    }
    u32 ta[4], qa[4], tsum = 0, qsum = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const bool live = (kb + (u64)e < b) && (lane * 4u + (u32)e < stop);
      ta[e] = live && !isi[e] ? len[e] : 0u;
      qa[e] = live && !isd[e] ? len[e] : 0u;
      tsum += ta[e];
      qsum += qa[e];
    }
    /* lane sums are < 2^30 and their wave prefix < 2^36: scan the two 16-bit halves (DPP) and recombine */
    u32 tl = 0, th = 0, ql = 0, qh = 0;
    u64 tp = 0, qp = 0;
    if (POS) {
      tl = wave_incl_scan_u32(tsum & 0xFFFFu), th = wave_incl_scan_u32(tsum >> 16);
      ql = wave_incl_scan_u32(qsum & 0xFFFFu), qh = wave_incl_scan_u32(qsum >> 16);
      const u64 t_incl = ((u64)th << 16) + (u64)tl, q_incl = ((u64)qh << 16) + (u64)ql;
      tp = t_base + t_incl - (u64)tsum, qp = q_base + q_incl - (u64)qsum; /* before this lane's first op */
    } else {
      acc_t += (u64)tsum;
      acc_q += (u64)qsum;
    }
    u32 prev = (u32)__shfl_up((int)code[3], 1u);
    if (lane == 0) prev = carry_code;
    u32 nxt = (u32)__shfl_down((int)code[0], 1u);
    if (lane == 63u) nxt = k0 + 256u < nops ? (rec[k0 + 256u] & 15u) : 0xFu; /* the record's next op, whoever walks it */
    bool is_ev[4];
    u32 nev = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const u32 pc = e == 0 ? prev : code[e - 1];
      const u32 nc = e == 3 ? nxt : code[e + 1];
      const bool live = (kb + (u64)e < b) && (lane * 4u + (u32)e < stop);
      const bool after_m = pc == WGA_OP_M || pc == WGA_OP_EQ || pc == WGA_OP_X;
      const bool cont_follows = nc == WGA_OP_I_CONT || nc == WGA_OP_D_CONT;
      const bool head_indel = code[e] == WGA_OP_I || code[e] == WGA_OP_D;
      is_ev[e] = live && ((code[e] == WGA_OP_X && snp) || (head_indel && after_m && ((u64)len[e] > svlen || cont_follows)));
      nev += is_ev[e] ? 1u : 0u;
    }
    if (!POS) acc_e += nev;
    const u32 einc = POS ? wave_incl_scan_u32(nev) : 0u;
    if (POS && eout && nev) {
      u64* e_out = eout + 3 * (e_base + (u64)(einc - nev));
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (is_ev[e]) {
          e_out[0] = kb + (u64)e;
          e_out[1] = tp;
          e_out[2] = qp;
          e_out += 3;
        }
        tp += ta[e];
        qp += qa[e];
      }
    }
    if (POS) {
      e_base += (u64)wave_last_u32(einc);
      t_base += ((u64)wave_last_u32(th) << 16) + (u64)wave_last_u32(tl);
      q_base += ((u64)wave_last_u32(qh) << 16) + (u64)wave_last_u32(ql);
    }
    carry_code = (u32)__shfl((int)code[3], 63);
    if (badm) {
      *bad_at = k0 + (u64)stop;
      break;
    }
  }
  if (!POS) { /* the lanes' sums, added up once */
    t_base += __shfl(wave_incl_scan_u64(acc_t, lane), 63);
    q_base += __shfl(wave_incl_scan_u64(acc_q, lane), 63);
    e_base += (u64)wave_sum_u32(acc_e);
  }
  tot->t = t_base - st.t;
  tot->q = q_base - st.q;
  tot->e = e_base - st.e;
}

/* one wave per record; records beyond `long_ops` ops are left to the piece kernels below (long_ops = 0: none are) */
__global__ __launch_bounds__(256) void k_paf_call_events(u32 n, const u32* __restrict__ ops,
                                                         const u64* __restrict__ op_off, u64 svlen,
                                                         u32 snp, u64* ev_cnt, u64* ev,
                                                         const u64* ev_off, u64 long_ops) {
  const u32 lane = threadIdx.x & 63u;
  const u64 i = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x);
  if (i >= n) return;
  const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
  if (long_ops && nops > long_ops) return;
  PafCallState z, tot;
  z.t = z.q = z.e = 0;
  u64 bad;
  if (ev) /* wave-uniform */
    paf_call_walk<true>(ops + o0, nops, 0, nops, svlen, snp, z, ev + 3 * ev_off[i], lane, &tot, &bad);
  else
    paf_call_walk<false>(ops + o0, nops, 0, nops, svlen, snp, z, (u64*)0, lane, &tot, &bad);
  if (lane == 0 && ev_cnt) ev_cnt[i] = tot.e;
}

/* ---- long records in pieces (the scheme of the MAF walks): a record beyond `long_ops` ops is cut into pieces of `piece_ops`
 *      (a multiple of 256), every piece is one wave's walk in a persistent grid over the piece list; a first walk leaves each
 *      piece's sums, one thread per record turns them into each piece's start state (running positions, events so far, "the
 *      walk has stopped": a piece behind the record's first bad op is dead), the second walk writes the events. ------------- */
__global__ __launch_bounds__(256) void k_op_piece_counts(u32 n, const u64* __restrict__ op_off, u64 long_ops, u64 piece_ops,
                                                         u32 all, u64* npieces) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 nops = op_off[i + 1] - op_off[i];
  npieces[i] = nops > long_ops ? (nops + piece_ops - 1) / piece_ops : (u64)all; /* all: the other records are one piece each */
}
struct wga_call_piece {
  u64 t, q, e;  /* MODE 0: the piece's sums; after the record scan: its start state */
  u64 bad;      /* MODE 0: record-relative first bad op or WGA_NONE; after the scan: 1 = dead, 0 = walk it */
};
/* a record's pieces are equal: its ops over its number of pieces, rounded up to whole 256-op steps (<= piece_ops) */
__device__ __forceinline__ u64 piece_span(u64 nops, u64 np) { return ((nops + np - 1) / np + 255u) & ~(u64)255u; }
/* the record of every piece (one thread per record): the walks read it instead of bisecting piece_off per piece */
__global__ __launch_bounds__(256) void k_op_piece_records(u32 n, const u64* __restrict__ piece_off, u32* piece_rec) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  for (u64 p = piece_off[i]; p < piece_off[i + 1]; p++) piece_rec[p] = i;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_paf_call_pieces(u32 n, const u32* __restrict__ ops, const u64* __restrict__ op_off,
                                                         u64 svlen, u32 snp, const u64* __restrict__ piece_off, const u32* __restrict__ piece_rec,
                                                         wga_call_piece* pc, u64* ev, const u64* ev_off) {
  const u32 lane = threadIdx.x & 63u;
  const u64 n_pieces = piece_off[n];
  const u64 n_waves = (u64)gridDim.x * 4u;
  for (u64 p = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x); p < n_pieces; p += n_waves) {
    const u32 i = WGA_UNI32(piece_rec[p]);
    const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
    const u64 psz = piece_span(nops, piece_off[i + 1] - piece_off[i]);
    const u64 a0 = (p - piece_off[i]) * psz, a = a0 < nops ? a0 : nops, b = a + psz < nops ? a + psz : nops;
    PafCallState st, tot;
    st.t = st.q = st.e = 0;
    u64 bad;
    if (MODE == 0) {
      paf_call_walk<false>(ops + o0, nops, a, b, svlen, snp, st, (u64*)0, lane, &tot, &bad);
      if (lane == 0) {
        wga_call_piece r;
        r.t = tot.t, r.q = tot.q, r.e = tot.e, r.bad = bad;
        pc[p] = r;
      }
    } else {
      const wga_call_piece r = pc[p];
      if (r.bad) continue; /* behind the record's first bad op: the reference's fold skips these ops */
      st.t = r.t, st.q = r.q, st.e = r.e;
      paf_call_walk<true>(ops + o0, nops, a, b, svlen, snp, st, ev + 3 * ev_off[i], lane, &tot, &bad);
    }
  }
}
/* one thread per long record: its pieces' sums -> start states; the record's event count */
__global__ __launch_bounds__(256) void k_paf_call_piece_scan(u32 n, const u64* __restrict__ piece_off, wga_call_piece* pc,
                                                             u64* ev_cnt) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 p0 = piece_off[i], p1 = piece_off[i + 1];
  if (p0 == p1) return;
  u64 t = 0, q = 0, e = 0;
  bool dead = false;
  for (u64 p = p0; p < p1; p++) {
    const wga_call_piece r = pc[p];
    wga_call_piece s;
    s.t = t, s.q = q, s.e = e, s.bad = dead ? 1u : 0u;
    pc[p] = s;
    if (!dead) {
      t += r.t, q += r.q, e += r.e;
      dead = r.bad != WGA_NONE;
    }
  }
  if (ev_cnt) ev_cnt[i] = e;
}

/* ============================================================================================ */
/* K8: CIGAR text -> packed ops on the device                                                   */
/* ============================================================================================ */
/* Same result as the host packer wga_cigar_pack (the nom tokeniser of cigar.rs:43-75 +
 * utils.rs:69-74), one wave per record.  Fast pass: the text goes through LDS 1 KiB at a time (16
 * bytes per lane); every non-digit byte is an op whose length is the digit run right before it,
 * parsed backwards from the staged text; op slots come from a wave scan of the per-lane op
 * counts.  Anything the fast pass does not cover — a token that is not "1..19 digits + one ASCII
 * char" (empty length, multi-char or multi-byte op, u64 overflow candidates), a length >= 2^28
 * (split into pieces), text that is empty or ends in digits — flags the record, and lane 0 then
 * re-tokenises it serially with the packer's exact rules and error reporting.  Records with
 * errors abort the run on the host anyway; the serial path only has to be right. */
struct wga_tok_err_dev {
  int err;      /* wga_rec_err */
  u32 tok_len;  /* offending token: length ... */
  u64 tok_off;  /* ... and offset inside the record's text */
};

__device__ __forceinline__ u32 tok_op_code(u8 c) {
  switch (c) {
    case 'M': return WGA_OP_M;
    case 'I': return WGA_OP_I;
    case 'D': return WGA_OP_D;
    case 'N': return WGA_OP_N;
    case 'S': return WGA_OP_S;
    case 'H': return WGA_OP_H;
    case 'P': return WGA_OP_P;
    case '=': return WGA_OP_EQ;
    case 'X': return WGA_OP_X;
    default: return WGA_OP_OTHER;
  }
}
__device__ __forceinline__ u32 tok_utf8_len(u8 c) {
  if (c < 0x80u) return 1u;
  if ((c >> 5) == 0x6u) return 2u;
  if ((c >> 4) == 0xEu) return 3u;
  if ((c >> 3) == 0x1Eu) return 4u;
  return 1u;
}

/* the packer's loop, verbatim in behaviour: returns the op count, writes ops if out != NULL */
__device__ __forceinline__ u64 tok_serial(const u8* text, u64 len, u32* out, wga_tok_err_dev* err) {
  u64 n = 0, p = 0;
  int e = WGA_REC_OK;
  u64 eoff = 0, elen = 0;
  if (len == 0) e = WGA_REC_PANIC;
  while (p < len) {
    const u64 ls = p;
    while (p < len && (u8)(text[p] - (u8)'0') < 10u) p++;
    const u64 ln = p - ls, os = p;
    while (p < len && (u8)(text[p] - (u8)'0') >= 10u) p++;
    const u64 on = p - os;
    if (on == 0 || tok_utf8_len(text[os]) < on) {
      e = WGA_REC_CIGAR_OP_INVALID;
      eoff = os;
      elen = on;
      break;
    }
    if (ln == 0) {
      e = WGA_REC_PARSE_INT;
      eoff = ls;
      elen = 0;
      break;
    }
    u64 v = 0;
    bool ovf = false;
    for (u64 k = 0; k < ln; k++) {
      const u64 d = (u64)(text[ls + k] - (u8)'0');
      if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10ull) {
        ovf = true;
        break;
      }
      v = v * 10ull + d;
    }
    if (ovf) {
      e = WGA_REC_PARSE_INT;
      eoff = ls;
      elen = ln;
      break;
    }
    const u32 code = on == 1 ? tok_op_code(text[os]) : (u32)WGA_OP_OTHER;
    const u32 cont = code == WGA_OP_I ? (u32)WGA_OP_I_CONT : code == WGA_OP_D ? (u32)WGA_OP_D_CONT : code;
    bool first = true;
    do {
      const u64 piece = v > (u64)WGA_OP_MAX_LEN ? (u64)WGA_OP_MAX_LEN : v;
      if (out) out[n] = ((u32)piece << 4) | (first ? code : cont);
      n++;
      v -= piece;
      first = false;
    } while (v > 0);
  }
  if (err) {
    err->err = e;
    err->tok_len = (u32)elen;
    err->tok_off = eoff;
  }
  return n;
}

#define WGA_TOK_HIST 32u /* bytes of the previous chunk kept in front of the staged one */

__global__ __launch_bounds__(256) void k_cigar_tokenise(u32 n, const u8* __restrict__ text,
                                                        const u64* __restrict__ text_beg,
                                                        const u64* __restrict__ text_end,
                                                        u64* op_cnt, wga_tok_err_dev* errs,
                                                        u32* ops, const u64* op_off) {
  __shared__ __attribute__((aligned(16))) u8 s_txt[4][WGA_TOK_HIST + 1024u + 16u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 i = (u64)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const u8* rec = text + text_beg[i]; /* CSR texts: text_end = text_beg + 1; spans of a file: two arrays */
  const u64 len = text_end[i] - text_beg[i];
  u32* out = ops ? ops + op_off[i] : (u32*)0;
  /* a record that turns out to hold an error packs to fewer ops than its non-digit bytes: the
   * fast pass must stay inside the record's own slots (the serial pass then rewrites them) */
  const u64 limit = ops ? op_off[i + 1] - op_off[i] : 0ull;
  u8* const buf = s_txt[wave];
  u8* const cur = buf + WGA_TOK_HIST; /* cur[-k] = byte k before the chunk */
  bool viol = len == 0 || (u8)(rec[len ? len - 1 : 0] - (u8)'0') < 10u; /* empty, or ends in digits */
  u64 base = 0;
  if (lane < WGA_TOK_HIST / 4u) ((u32*)buf)[lane] = 0x30303030u; /* history before the text: all '0' is never read
                                                                    as part of a run because p == 0 is checked */
  for (u64 c0 = 0; c0 < len && !viol; c0 += 1024) {
    WGA_WAVE_SYNC();
    const u64 c = c0 + (u64)lane * 16u;
    const u32 nv = c >= len ? 0u : (len - c >= 16u ? 16u : (u32)(len - c));
    u32 w[4] = {0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u};
    if (nv == 16u) {
      const u32x4_a1 a = *(const u32x4_a1*)(rec + c);
      w[0] = a[0], w[1] = a[1], w[2] = a[2], w[3] = a[3];
    } else if (nv) {
      for (u32 j = 0; j < nv; j++) {
        w[j >> 2] &= ~(0xFFu << (8u * (j & 3u)));
        w[j >> 2] |= (u32)rec[c + j] << (8u * (j & 3u));
      }
    }
    {
      const u32x4_a16 wv = {w[0], w[1], w[2], w[3]};
      *(u32x4_a16*)(cur + lane * 16u) = wv;
    }
    WGA_WAVE_SYNC();
    /* non-digit bytes of this lane (padding beyond the text counts as digits) */
    u32 nd = 0; /* 16-bit mask */
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const u32 ch = (w[d] >> (8 * b)) & 0xFFu;
        nd |= ((ch - 0x30u) >= 10u ? 1u : 0u) << (4 * d + b);
      }
    }
    const u32 cnt = (u32)__builtin_popcount(nd);
    const u32 incl = wave_incl_scan_u32(cnt);
    u32 slot = incl - cnt;
    u32 m = nd;
    bool bad = false;
    while (m) {
      const u32 j = (u32)__builtin_ctz(m);
      m &= m - 1u;
      const int p = (int)(lane * 16u + j); /* position inside the chunk */
      const u8 ch = cur[p];
      /* digits right before p, backwards (at most 19 accepted) */
      u64 v = 0, mul = 1;
      u32 nd_run = 0;
      int k = p - 1;
      const int kmin = -(int)(c0 < (u64)WGA_TOK_HIST ? c0 : (u64)WGA_TOK_HIST); /* readable history */
      while (k >= kmin && (u8)(cur[k] - (u8)'0') < 10u && nd_run < 20u) {
        v += (u64)(cur[k] - (u8)'0') * mul;
        mul *= 10ull;
        nd_run++;
        k--;
      }
      /* not "1..19 digits + one ASCII char", or the run may go on beyond what is staged, or the
       * length needs splitting: leave the record to the serial path */
      if (nd_run == 0u || nd_run > 19u || ch >= 0x80u || (k < kmin && c0 + (u64)p > (u64)nd_run) ||
          v > (u64)WGA_OP_MAX_LEN)
        bad = true;
      if (out && !bad && base + slot < limit) out[base + slot] = ((u32)v << 4) | tok_op_code(ch);
      slot++;
    }
    viol = __ballot(bad) != 0ull;
    base += (u64)wave_last_u32(incl);
    /* keep the chunk's last WGA_TOK_HIST bytes in front of the next one */
    WGA_WAVE_SYNC();
    u32 hv = 0;
    if (lane < WGA_TOK_HIST / 4u) hv = ((const u32*)(cur + 1024u - WGA_TOK_HIST))[lane];
    WGA_WAVE_SYNC();
    if (lane < WGA_TOK_HIST / 4u) ((u32*)buf)[lane] = hv;
  }
  wga_tok_err_dev e;
  e.err = WGA_REC_OK;
  e.tok_len = 0;
  e.tok_off = 0;
  if (viol) { /* wave-uniform */
    if (lane == 0) base = tok_serial(rec, len, out, &e);
  }
  if (lane == 0) {
    if (op_cnt) op_cnt[i] = base;
    if (errs) errs[i] = e;
  }
}

/* ============================================================================================ */
/* K9: pafcov BED text                                                                          */
/* ============================================================================================ */
/* pafcov prints one line per target base, "<name>\t<pos>\t<pos+1>\t<count>\n" (pafcov.rs:56-60):
 * pure formatting, and the bulk of the tool's wall time.  Line lengths are a function of the
 * position and the count (scan functor), then one thread writes one line. */
__device__ __forceinline__ u32 dec_digits(u64 v) {
  u32 n = 1;
  if (v >= 10000000000ull) {
    v /= 10000000000ull;
    n += 10;
  }
  u32 w = (u32)v; /* < 10^10 does not fit u32 entirely: handle the top digit */
  if (v >= 1000000000ull) return n + 9u;
  if (w >= 100000000u) return n + 8u;
  if (w >= 10000000u) return n + 7u;
  if (w >= 1000000u) return n + 6u;
  if (w >= 100000u) return n + 5u;
  if (w >= 10000u) return n + 4u;
  if (w >= 1000u) return n + 3u;
  if (w >= 100u) return n + 2u;
  if (w >= 10u) return n + 1u;
  return n;
}
/* writes the decimal digits of v (nd = dec_digits(v)) at p[0 .. nd) */
__device__ __forceinline__ void dec_write(u8* p, u64 v, u32 nd) {
  if (v < 0x100000000ull) {
    u32 w = (u32)v;
    for (u32 k = nd; k-- > 0;) {
      p[k] = (u8)('0' + w % 10u);
      w /= 10u;
    }
  } else {
    for (u32 k = nd; k-- > 0;) {
      p[k] = (u8)('0' + (u32)(v % 10ull));
      v /= 10ull;
    }
  }
}
struct ScanCovLine {
  const int* cov;
  u64 p0;
  u32 name_len;
  __device__ u64 operator()(u32 i) const {
    const u64 p = p0 + i;
    return (u64)name_len + 4ull + dec_digits(p) + dec_digits(p + 1) + dec_digits((u64)(u32)cov[i]);
  }
};
__global__ __launch_bounds__(256) void k_pafcov_format(ScanCovLine f, u32 n, const u8* __restrict__ name,
                                                       const u64* __restrict__ line_off,
                                                       u8* __restrict__ out) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  u8* p = out + (line_off[i] - line_off[0]);
  for (u32 k = 0; k < f.name_len; k++) p[k] = name[k];
  p += f.name_len;
  const u64 pos = f.p0 + i;
  const u32 d0 = dec_digits(pos), d1 = dec_digits(pos + 1);
  const u64 c = (u64)(u32)f.cov[i];
  const u32 d2 = dec_digits(c);
  *p++ = (u8)'\t';
  dec_write(p, pos, d0);
  p += d0;
  *p++ = (u8)'\t';
  dec_write(p, pos + 1, d1);
  p += d1;
  *p++ = (u8)'\t';
  dec_write(p, c, d2);
  p += d2;
  *p = (u8)'\n';
}

/* ============================================================================================ */
/* K10: paf2chain data lines                                              */
/* ============================================================================================ */
/* parse_cigar_to_chain + cigar_unit_chain (cigar.rs:251-295,460-490): runs of M / = / X ops form a
 * block; when an M-like op follows an indel group and a block is open, the line
 * "\n<size>\t<D bases of the group>\t<I bases of the group>" goes out; leading indels are dropped,
 * the last block ends the record as "\n<size>" (trailing indels dropped).  parse_cigar_to_trim
 * (cigar.rs:202-245) for the chain header: I / D bases before the first M-like op, and the
 * length of the LAST I / D op behind the last M-like op (assignment, not a sum).
 * One wave per record, 4 consecutive ops per lane and 256 per step.  With exclusive prefix sums
 * PM, PD, PI of the M-like / D / I lengths, the line raised at op j is the difference between the
 * prefix triple at j and the triple at the previous raising op (the first M-like op for the first
 * line): the triples of a step go through LDS so that every raising op can read its predecessor.
 * Two passes: text bytes per record, then the text. */
struct wga_chain_trim {
  u64 head_ins, head_del, tail_ins, tail_del;
};

/* the reference's loops verbatim on packed ops (lane 0 only): used for records with zero-length ops,
 * whose "size != 0" / "diffs != 0" tests (cigar.rs:472) the prefix formulation does not cover */
__device__ __forceinline__ void chain_serial(const u32* rec, u64 nops, u8* text, wga_chain_trim& tr,
                                             u64& nbytes, u64& bad_idx) {
  u64 size = 0, qd = 0, td = 0, off = 0;
  u64 head_ins = 0, head_del = 0, tail_ins = 0, tail_del = 0;
  bool head = true;
  bad_idx = WGA_NONE;
  for (u64 k = 0; k < nops; k++) {
    const u32 code = rec[k] & 15u;
    const u64 len = rec[k] >> 4;
    if (code == WGA_OP_M || code == WGA_OP_EQ || code == WGA_OP_X) {
      if (size != 0 && td + qd != 0) {
        const u32 a = dec_digits(size), b = dec_digits(qd), c = dec_digits(td);
        if (text) {
          u8* p = text + off;
          *p++ = (u8)'\n';
          dec_write(p, size, a);
          p += a;
          *p++ = (u8)'\t';
          dec_write(p, qd, b);
          p += b;
          *p++ = (u8)'\t';
          dec_write(p, td, c);
        }
        off += 3u + a + b + c;
        size = 0;
      }
      size += len;
      td = qd = 0;
      tail_ins = tail_del = 0;
      head = false;
    } else if (code == WGA_OP_I || code == WGA_OP_I_CONT) {
      td += len;
      if (head) head_ins += len;
      tail_ins = code == WGA_OP_I ? len : tail_ins + len;
    } else if (code == WGA_OP_D || code == WGA_OP_D_CONT) {
      qd += len;
      if (head) head_del += len;
      tail_del = code == WGA_OP_D ? len : tail_del + len;
    } else {
      bad_idx = k;
      break;
    }
  }
  const u32 dl = dec_digits(size);
  if (text && bad_idx == WGA_NONE) {
    u8* p = text + off;
    *p++ = (u8)'\n';
    dec_write(p, size, dl);
  }
  nbytes = off + 1u + dl;
  tr.head_ins = head_ins;
  tr.head_del = head_del;
  tr.tail_ins = tail_ins;
  tr.tail_del = tail_del;
}

/* ---- the wave-parallel walk ------------------------------------------------------------------
 * A step is WGA_CHAIN_STEP = 64 x WGA_CHAIN_OPL consecutive ops, WGA_CHAIN_OPL consecutive ops per lane, kept in
 * registers.  An op RAISES a line when it is M-like and the op in front of it is not (the record's first op never
 * raises).  Per lane, one unrolled pass over its ops runs the reference's accumulators (size, D bases, I bases since
 * the last raise) and stores their value at every raise into the wave's slot list in LDS, at the rank a wave scan of
 * the raise counts gives; what is open at the lane's end goes through three wave scans so that every lane learns what
 * was open at its start (from the nearest lower lane that raised, or from the previous step), and adds that to its
 * first line.  The lines are then formatted DENSELY, one line per lane and 64 per round — digit counts in the count
 * pass; in the fill pass the text of a round is put together in an LDS buffer that mirrors the output's position
 * inside its 128-byte line, and only whole lines are stored (16 bytes per lane; the rest waits for the next round):
 * no line of the output is written in pieces.
 * The state between steps is the reference's own (size, D bases, I bases of its loop, cigar.rs:460-490), and the
 * reference's tests are on VALUES ("size != 0", "diffs != 0", cigar.rs:472): with lengths >= 1 they say "M-like op behind
 * an indel op, and not the record's first M-like op", which is what the lane pass finds by op class (a line whose size
 * is 0 — the first M-like op behind leading indels — is not printed; its D / I sums are the head trim).  A step that
 * holds a zero-length op or an op outside M = X I D (a wave vote) runs the reference's loop as it stands over its ops —
 * scalar, the words read out of the lanes' registers — and hands its lines to the same formatting rounds; the fold
 * ends at the op outside M = X I D.  Everything is u32 and exact as long as the open sums plus a step's lengths stay
 * below 2^32 (voted per step); a record that passes that goes through chain_serial as a whole.  The tail trim is
 * read off the record's last ops afterwards. */
#ifndef WGA_CHAIN_OPL
#define WGA_CHAIN_OPL 8
#endif
#define WGA_CHAIN_STEP (64u * WGA_CHAIN_OPL)
#define WGA_CHAIN_NL (32u * WGA_CHAIN_OPL) /* at most every other op raises */
#define WGA_CHAIN_TB 2432u                 /* < 128 carried bytes + 64 lines x 33 bytes + the last line, + 128 read ahead */
#define WGA_CHAIN_MM ((1u << WGA_OP_M) | (1u << WGA_OP_EQ) | (1u << WGA_OP_X))
#define WGA_CHAIN_IM ((1u << WGA_OP_I) | (1u << WGA_OP_I_CONT))
#define WGA_CHAIN_DM ((1u << WGA_OP_D) | (1u << WGA_OP_D_CONT))
#define WGA_CHAIN_PAD 0xFu /* behind the record's end: no length, no class */

/* decimal digits of a u32; p10[t] = 10^t, t < 10 */
__device__ __forceinline__ u32 dec_digits_u32(u32 v, const u32* p10) {
  const u32 x = v | 1u;
  const u32 t = ((32u - (u32)__clz(x)) * 1233u) >> 12;
  return t + 1u - (x < p10[t] ? 1u : 0u);
}

struct ChainWalk {
  u32 c_size, c_qd, c_td; /* the reference's size / query_diff / target_diff in front of the step (wave-uniform) */
  u32 seen_m;             /* an M-like op was seen: the head trim is closed */
  u32 head_ins, head_del;
  u64 nbytes;             /* count pass: this lane's share of the text bytes */
  u32 fill, head_skip;    /* fill pass: bytes in the text buffer; those in front of head_skip are not this record's */
  u8* gpos;               /* fill pass: where byte 0 of the text buffer belongs (128-byte aligned) */
};

__device__ __forceinline__ void chain_load(const u32* rec, u64 nops, u64 k0, u32 lane, u32 (&w)[WGA_CHAIN_OPL]) {
  const u64 kb = k0 + (u64)lane * WGA_CHAIN_OPL;
  if (kb + WGA_CHAIN_OPL <= nops) {
#pragma unroll
    for (int j = 0; j < (int)WGA_CHAIN_OPL / 4; j++) {
      const u32x4_a4 v = *(const u32x4_a4*)(rec + kb + 4 * j);
      w[4 * j] = v[0], w[4 * j + 1] = v[1], w[4 * j + 2] = v[2], w[4 * j + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) w[e] = kb + (u64)e < nops ? rec[kb + e] : WGA_CHAIN_PAD;
  }
}

/* whole 128-byte lines of the text buffer go out, the rest moves to the buffer's start */
__device__ __forceinline__ void chain_flush_lines(ChainWalk& st, u8* tbuf, u32 lane) {
  const u32 nfull = st.fill >> 7;
  if (nfull == 0u) return; /* wave-uniform */
  WGA_WAVE_SYNC();
  for (u32 g = lane; g < nfull * 8u; g += 64u) {
    const u32 bo = g * 16u;
    if (bo >= st.head_skip) {
      *(u32x4_a16*)(st.gpos + bo) = *(const u32x4_a16*)(tbuf + bo);
    } else if (bo + 16u > st.head_skip) {
      for (u32 k = st.head_skip; k < bo + 16u; k++) st.gpos[k] = tbuf[k];
    }
  }
  const u32 keep = lane < 32u ? *(const u32*)(tbuf + nfull * 128u + lane * 4u) : 0u;
  WGA_WAVE_SYNC();
  if (lane < 32u) *(u32*)(tbuf + lane * 4u) = keep;
  WGA_WAVE_SYNC();
  st.gpos += (u64)nfull * 128u;
  st.fill &= 127u;
  st.head_skip = 0u;
}

/* v's decimal digits end in front of tbuf[end]; returns where they start */
__device__ __forceinline__ u32 chain_put_dec(u8* tbuf, u32 end, u32 v) {
  do {
    const u32 q = v / 10u;
    tbuf[--end] = (u8)('0' + (v - q * 10u));
    v = q;
  } while (v);
  return end;
}

/* the lines of a step by op class, 8 ops per lane: returns their number, leaves them in slot[] */
__device__ __forceinline__ u32 chain_find_lines(const u32 (&w)[WGA_CHAIN_OPL], u32 mb, u32 lane, u32x4_a16* slot,
                                                ChainWalk& st) {
  u32 pm = (u32)__shfl_up((int)(mb >> (WGA_CHAIN_OPL - 1u)), 1u);
  if (lane == 0u) pm = (st.c_qd | st.c_td) == 0u ? 1u : 0u; /* no indel open: the op in front is M-like, or none */
  const u32 rb = mb & ~((mb << 1) | (pm & 1u));
  const u32 cnt = (u32)__popc(rb);
  const u32 einc = wave_incl_scan_u32(cnt);
  const u32 nl = WGA_UNI32(wave_last_u32(einc));
  /* the accumulators; a raise stores them and starts them again */
  u32 pos = einc - cnt, size = 0, qd = 0, td = 0;
#pragma unroll
  for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) {
    const u32 code = w[e] & 15u, len = w[e] >> 4;
    const u32 r = bit_mask(rb, (u32)e);
    if (r) {
      u32x4_a16 v;
      v[0] = size, v[1] = qd, v[2] = td, v[3] = 0u;
      slot[pos] = v;
    }
    pos -= r;
    size = (size & ~r) + (len & bit_mask(WGA_CHAIN_MM, code));
    qd = (qd & ~r) + (len & bit_mask(WGA_CHAIN_DM, code));
    td = (td & ~r) + (len & bit_mask(WGA_CHAIN_IM, code));
  }
  /* what was open at the lane's start */
  const u32 s0 = wave_incl_scan_u32(size), s1 = wave_incl_scan_u32(qd), s2 = wave_incl_scan_u32(td);
  const u32 x0 = s0 - size, x1 = s1 - qd, x2 = s2 - td;
  const u64 hb = __ballot(cnt != 0u);
  const u64 below = hb & ((1ull << lane) - 1ull);
  const int src = below ? 63 - (int)__builtin_clzll(below) : 0;
  const u32 y0 = (u32)__shfl((int)x0, src), y1 = (u32)__shfl((int)x1, src), y2 = (u32)__shfl((int)x2, src);
  if (cnt) {
    u32x4_a16 v = slot[einc - cnt];
    v[0] += below ? x0 - y0 : st.c_size + x0;
    v[1] += below ? x1 - y1 : st.c_qd + x1;
    v[2] += below ? x2 - y2 : st.c_td + x2;
    slot[einc - cnt] = v;
  }
  const int last = hb ? 63 - (int)__builtin_clzll(hb) : 0;
  const u32 z0 = (u32)__shfl((int)x0, last), z1 = (u32)__shfl((int)x1, last), z2 = (u32)__shfl((int)x2, last);
  const u32 t0 = wave_last_u32(s0), t1 = wave_last_u32(s1), t2 = wave_last_u32(s2);
  st.c_size = WGA_UNI32(hb ? t0 - z0 : st.c_size + t0);
  st.c_qd = WGA_UNI32(hb ? t1 - z1 : st.c_qd + t1);
  st.c_td = WGA_UNI32(hb ? t2 - z2 : st.c_td + t2);
  return nl;
}

/* the same by the reference's loop (cigar.rs:460-490 + the head of 202-245), one op after the other with wave-uniform
 * values: for steps that hold zero-length ops or ops outside M = X I D.  nops_step: the ops of the step that exist.
 * Returns the number of lines; *stop = index (inside the step) of the first op outside M = X I D, or ~0. */
__device__ __forceinline__ u32 chain_find_lines_serial(const u32 (&w)[WGA_CHAIN_OPL], u32 nops_step, u32 lane,
                                                       u32x4_a16* slot, ChainWalk& st, u32* stop) {
  u32 size = st.c_size, qd = st.c_qd, td = st.c_td, nl = 0;
  *stop = 0xFFFFFFFFu;
  for (u32 l = 0; l * WGA_CHAIN_OPL < nops_step; l++) {
    bool out = false;
#pragma unroll
    for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) {
      const u32 x = l * WGA_CHAIN_OPL + (u32)e;
      const u32 word = WGA_UNI32((u32)__shfl((int)w[e], (int)l));
      if (out || x >= nops_step) continue;
      const u32 code = word & 15u, len = word >> 4;
      if ((WGA_CHAIN_MM >> code) & 1u) {
        if (size != 0u && (qd | td) != 0u) {
          if (lane == 0u) {
            u32x4_a16 v;
            v[0] = size, v[1] = qd, v[2] = td, v[3] = 0u;
            slot[nl] = v;
          }
          nl++;
          size = 0u;
        }
        if (!st.seen_m) st.head_del = qd, st.head_ins = td, st.seen_m = 1u;
        size += len;
        qd = td = 0u;
      } else if ((WGA_CHAIN_IM >> code) & 1u) {
        td += len;
      } else if ((WGA_CHAIN_DM >> code) & 1u) {
        qd += len;
      } else {
        *stop = x;
        out = true;
      }
    }
    if (out) break;
  }
  st.c_size = size, st.c_qd = qd, st.c_td = td;
  return nl;
}

/* one step: 0 = go on, 1 = the fold ended at an op outside M = X I D (*stop), 2 = sums past 2^32: chain_serial */
template <bool FILL, bool TAIL>
__device__ __forceinline__ int chain_step(const u32 (&w)[WGA_CHAIN_OPL], u32 nops_step, u32 lane, u32x4_a16* slot,
                                          const u32* p10, u8* tbuf, ChainWalk& st, u32* stop) {
  /* classes as bit masks, the checks, the lane's total length */
  const u32 lo = lane * WGA_CHAIN_OPL;
  u32 mb = 0, seen = 0, minw = 0xFFFFFFFFu, tl = 0;
#pragma unroll
  for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) {
    const u32 wc = TAIL ? (lo + (u32)e < nops_step ? w[e] : (1u << 4 | WGA_OP_EQ)) : w[e];
    mb |= ((WGA_CHAIN_MM >> (w[e] & 15u)) & 1u) << e;
    seen |= 1u << (wc & 15u);
    minw = wc < minw ? wc : minw;
    tl += w[e] >> 4;
  }
  if ((u64)st.c_size + (u64)st.c_qd + (u64)st.c_td + wave_sum_u32_wide(tl) > 0xFFFFFFFFull) return 2;
  const bool odd = (seen & ~(WGA_CHAIN_MM | WGA_CHAIN_IM | WGA_CHAIN_DM)) != 0u || minw < 16u;
  const bool head_open = st.seen_m == 0u;
  u32 nl;
  *stop = 0xFFFFFFFFu;
  if (__ballot(odd)) { /* wave-uniform */
    nl = chain_find_lines_serial(w, nops_step, lane, slot, st, stop);
  } else {
    nl = chain_find_lines(w, mb, lane, slot, st);
    if (head_open && __ballot(mb != 0u)) st.seen_m = 1u; /* the head trim: line 0's sums when its size is 0, else 0 */
  }
  if (nl) { /* wave-uniform */
    WGA_WAVE_SYNC();
    /* the lines, 64 per round; one whose size is 0 is the first M-like op behind leading indels: not printed */
    for (u32 base = 0; base < nl; base += 64u) {
      const u32 j = base + lane;
      u32x4_a16 v;
      v[0] = v[1] = v[2] = v[3] = 0u;
      if (j < nl) v = slot[j];
      if (head_open && base == 0u) {
        const u32 hs = (u32)__shfl((int)v[0], 0), hd = (u32)__shfl((int)v[1], 0), hi = (u32)__shfl((int)v[2], 0);
        if (hs == 0u) st.head_del = hd, st.head_ins = hi;
      }
      const bool on = j < nl && v[0] != 0u;
      const u32 d0 = dec_digits_u32(v[0], p10), d1 = dec_digits_u32(v[1], p10), d2 = dec_digits_u32(v[2], p10);
      const u32 ll = on ? 3u + d0 + d1 + d2 : 0u;
      if (!FILL) {
        st.nbytes += ll;
      } else {
        const u32 linc = wave_incl_scan_u32(ll);
        if (on) {
          u32 p = st.fill + linc; /* the line's end */
          p = chain_put_dec(tbuf, p, v[2]);
          tbuf[--p] = (u8)'\t';
          p = chain_put_dec(tbuf, p, v[1]);
          tbuf[--p] = (u8)'\t';
          p = chain_put_dec(tbuf, p, v[0]);
          tbuf[--p] = (u8)'\n';
        }
        st.fill += WGA_UNI32(wave_last_u32(linc));
        chain_flush_lines(st, tbuf, lane);
      }
    }
    WGA_WAVE_SYNC(); /* the slots are rewritten by the next step */
  }
  return *stop != 0xFFFFFFFFu ? 1 : 0;
}

/* parse_cigar_to_trim's tail (cigar.rs:202-245): the length of the last I (D) op behind the last M-like op, a split
 * length being its head op plus the continuation pieces — the I-class (D-class) lengths from the last op that is
 * M-like or an I (D) head on.  Clean records only; read backwards, 64 ops per look. */
__device__ __forceinline__ void chain_tail_trim(const u32* rec, u64 nops, u32 lane, u64& tail_ins, u64& tail_del) {
  tail_ins = tail_del = 0;
  bool done_i = false, done_d = false;
  for (u64 hi = nops; hi > 0 && !(done_i && done_d);) {
    const u64 lo = hi > 64u ? hi - 64u : 0u;
    const bool in = lo + lane < hi;
    const u32 w = in ? rec[lo + lane] : WGA_CHAIN_PAD;
    const u32 code = w & 15u, len = w >> 4;
    const bool m = ((WGA_CHAIN_MM >> code) & 1u) != 0u;
    const bool ic = ((WGA_CHAIN_IM >> code) & 1u) != 0u, dc = ((WGA_CHAIN_DM >> code) & 1u) != 0u;
    if (!done_i) {
      const u64 stop = __ballot(m || code == WGA_OP_I);
      const u32 from = stop ? 63u - (u32)__builtin_clzll(stop) : 0u;
      tail_ins += wave_sum_u64(ic && lane >= from ? (u64)len : 0ull);
      done_i = stop != 0ull;
    }
    if (!done_d) {
      const u64 stop = __ballot(m || code == WGA_OP_D);
      const u32 from = stop ? 63u - (u32)__builtin_clzll(stop) : 0u;
      tail_del += wave_sum_u64(dc && lane >= from ? (u64)len : 0ull);
      done_d = stop != 0ull;
    }
    hi = lo;
  }
}

/* One wave walks `nops` ops from `rec` as a record of its own: `first` — the head trim is open (the record starts here);
 * `last` — the walk ends the record ("\n<size>", trailing indels dropped), otherwise an M-like op follows an open block
 * and an open indel group (a cut of chain_find_cut) and what is open goes out as a line.  Count pass: returns the text bytes;
 * fill pass: writes them at `text`.  *weird: sums past 2^32 (nothing usable was produced); *bad_idx: first op outside
 * M = X I D, relative to rec. */
template <bool FILL>
__device__ __forceinline__ u64 chain_walk(const u32* rec, u64 nops, u8* text, bool first, bool last, u32 lane,
                                          u32x4_a16* slot, const u32* p10, u8* tbuf, ChainWalk& st, bool* weird,
                                          u64* bad_idx) {
  st.c_size = st.c_qd = st.c_td = 0u;
  st.seen_m = first ? 0u : 1u;
  st.head_ins = st.head_del = 0u;
  st.nbytes = 0;
  st.fill = st.head_skip = FILL ? (u32)((uintptr_t)text & 127u) : 0u;
  st.gpos = FILL ? text - st.fill : (u8*)0;
  *weird = false;
  *bad_idx = WGA_NONE;
  u32 w[WGA_CHAIN_OPL];
  if (nops) chain_load(rec, nops, 0, lane, w);
  for (u64 k0 = 0; k0 < nops; k0 += WGA_CHAIN_STEP) {
    u32 nw[WGA_CHAIN_OPL];
    const bool more = k0 + WGA_CHAIN_STEP < nops;
    if (more) chain_load(rec, nops, k0 + WGA_CHAIN_STEP, lane, nw);
    u32 stop;
    const int rc = more || k0 + WGA_CHAIN_STEP == nops
                       ? chain_step<FILL, false>(w, WGA_CHAIN_STEP, lane, slot, p10, tbuf, st, &stop)
                       : chain_step<FILL, true>(w, (u32)(nops - k0), lane, slot, p10, tbuf, st, &stop);
    if (rc == 2) { /* wave-uniform */
      *weird = true;
      return 0;
    }
    if (rc == 1) {
      *bad_idx = k0 + (u64)stop;
      break;
    }
    if (more) {
#pragma unroll
      for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) w[e] = nw[e];
    }
  }
  /* the last block: "\n<size>" (cigar.rs:289-291; 0 when the record has no M-like op); at a cut: the line the next op raises */
  const u32 d0 = dec_digits_u32(st.c_size, p10);
  const u32 el = last ? 1u + d0 : 3u + d0 + dec_digits_u32(st.c_qd, p10) + dec_digits_u32(st.c_td, p10);
  if (FILL) {
    if (lane == 0) {
      u32 p = st.fill + el;
      if (!last) {
        p = chain_put_dec(tbuf, p, st.c_td);
        tbuf[--p] = (u8)'\t';
        p = chain_put_dec(tbuf, p, st.c_qd);
        tbuf[--p] = (u8)'\t';
      }
      p = chain_put_dec(tbuf, p, st.c_size);
      tbuf[p - 1u] = (u8)'\n';
    }
    st.fill += el;
    chain_flush_lines(st, tbuf, lane);
    WGA_WAVE_SYNC();
    for (u32 k = st.head_skip + lane; k < st.fill; k += 64u) st.gpos[k] = tbuf[k];
    return 0;
  }
  return wave_sum_u64(st.nbytes) + el;
}

/* one wave per record; records beyond `long_ops` ops are left to the piece kernels below (long_ops = 0: none are) */
template <bool FILL>
__global__ __launch_bounds__(256) void k_cigar_chain(u32 n, const u32* __restrict__ ops,
                                                     const u64* __restrict__ op_off,
                                                     wga_chain_trim* trims, u64* nbytes,
                                                     wga_rec_diag* diag, u8* out,
                                                     const u64* out_off, u64 long_ops) {
  __shared__ u32x4_a16 s_slot[4][WGA_CHAIN_NL];
  __shared__ u32 s_p10[4][16];
  __shared__ u32x4_a16 s_text[4][FILL ? WGA_CHAIN_TB / 16u : 1u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 i = (u64)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
  if (long_ops && nops > long_ops) return;
  const u32* rec = ops + o0;
  u8* const text = FILL ? out + out_off[i] : (u8*)0;
  u32x4_a16* const slot = s_slot[wave];
  u32* const p10 = s_p10[wave];
  u8* const tbuf = (u8*)s_text[wave];
  if (lane < 10u) {
    u32 v = 1u;
    for (u32 k = 0; k < lane; k++) v *= 10u;
    p10[lane] = v;
  }
  WGA_WAVE_SYNC();
  ChainWalk st;
  bool weird;
  u64 bad_idx;
  const u64 nb = chain_walk<FILL>(rec, nops, text, true, true, lane, slot, p10, tbuf, st, &weird, &bad_idx);
  if (weird) { /* a block, or an indel group, of 2^32 bases or more: the reference's loop as it stands, in u64 */
    if (lane == 0) {
      wga_chain_trim tr;
      u64 nbs = 0, bad = WGA_NONE;
      chain_serial(rec, nops, text, tr, nbs, bad);
      if (!FILL) {
        nbytes[i] = nbs;
        trims[i] = tr;
        if (bad != WGA_NONE) diag[i].bad_op_idx = bad;
      }
    }
    return;
  }
  if (!FILL) {
    wga_chain_trim tr;
    /* indels in front of the first M-like op; everything when no M-like op exists */
    tr.head_ins = st.seen_m ? (u64)st.head_ins : (u64)st.c_td;
    tr.head_del = st.seen_m ? (u64)st.head_del : (u64)st.c_qd;
    chain_tail_trim(rec, nops, lane, tr.tail_ins, tr.tail_del);
    if (lane == 0) {
      nbytes[i] = nb;
      trims[i] = tr;
      if (bad_idx != WGA_NONE) diag[i].bad_op_idx = bad_idx;
    }
  }
}

/* ---- long records in pieces: the fold's state is (size, D bases, I bases) and it starts again at every line, so a record
 *      can be cut where a line is certain: op k is M-like, op k-1 an indel op of length >= 1, op k-2 an M-like op of length
 *      >= 1 ("size != 0 && diffs != 0" holds at k whatever came before, cigar.rs:472).  Piece j of a record beyond
 *      `long_ops` ops starts at the first such k at or behind j * piece_span (none inside its span: the piece is
 *      empty and its ops stay with the piece in front) and ends where the next non-empty piece starts; it is walked as a
 *      record of its own (chain_walk) whose last line is the one op k raises.  Count walk: bytes per piece; one thread per
 *      record: the record's bytes and every piece's place in its text; fill walk: the text.  A piece whose sums pass 2^32
 *      sends its record through chain_serial (by the scan thread, then by piece 0's lane 0). ------------------------------ */
struct wga_chain_piece {
  u64 a, b;   /* ops [a, b) of the record; a = WGA_NONE: empty */
  u64 nb;     /* count walk: text bytes (WGA_NONE: sums past 2^32) */
  u64 off;    /* record scan: the piece's text starts here inside the record's (WGA_NONE: the record is chain_serial's) */
};
__device__ __forceinline__ u64 chain_find_cut(const u32* rec, u64 lo, u64 hi, u32 lane) {
  for (u64 base = lo < 2u ? 2u : lo; base < hi; base += 64u) {
    const u64 k = base + lane;
    bool ok = false;
    if (k < hi) {
      const u32 w0 = rec[k], w1 = rec[k - 1], w2 = rec[k - 2];
      ok = ((WGA_CHAIN_MM >> (w0 & 15u)) & 1u) && (((WGA_CHAIN_IM | WGA_CHAIN_DM) >> (w1 & 15u)) & 1u) && (w1 >> 4) != 0u &&
           ((WGA_CHAIN_MM >> (w2 & 15u)) & 1u) && (w2 >> 4) != 0u;
    }
    const u64 m = __ballot(ok);
    if (m) return base + (u64)(__ffsll((unsigned long long)m) - 1);
  }
  return WGA_NONE;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_cigar_chain_pieces(u32 n, const u32* __restrict__ ops, const u64* __restrict__ op_off,
                                                            const u64* __restrict__ piece_off, const u32* __restrict__ piece_rec,
                                                            wga_chain_piece* pc, wga_chain_trim* trims, wga_rec_diag* diag,
                                                            u8* out, const u64* out_off) {
  __shared__ u32x4_a16 s_slot[4][WGA_CHAIN_NL];
  __shared__ u32 s_p10[4][16];
  __shared__ u32x4_a16 s_text[4][MODE ? WGA_CHAIN_TB / 16u : 1u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  u32x4_a16* const slot = s_slot[wave];
  u32* const p10 = s_p10[wave];
  u8* const tbuf = (u8*)s_text[wave];
  if (lane < 10u) {
    u32 v = 1u;
    for (u32 k = 0; k < lane; k++) v *= 10u;
    p10[lane] = v;
  }
  WGA_WAVE_SYNC();
  const u64 n_pieces = piece_off[n];
  const u64 n_waves = (u64)gridDim.x * 4u;
  for (u64 p = (u64)blockIdx.x * 4 + wave; p < n_pieces; p += n_waves) {
    const u32 i = WGA_UNI32(piece_rec[p]);
    const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
    const u32* rec = ops + o0;
    const u64 j = p - piece_off[i], np = piece_off[i + 1] - piece_off[i];
    ChainWalk st;
    bool weird;
    u64 bad;
    if (MODE == 0) {
      wga_chain_piece r;
      r.nb = 0, r.off = 0;
      const u64 psz = piece_span(nops, np);
      const u64 lo = j * psz;
      r.a = j == 0 ? 0 : chain_find_cut(rec, lo, lo + psz < nops ? lo + psz : nops, lane);
      r.b = WGA_NONE;
      if (r.a != WGA_NONE) {
        for (u64 jj = j + 1; jj < np && r.b == WGA_NONE; jj++) {
          const u64 l2 = jj * psz;
          r.b = chain_find_cut(rec, l2, l2 + psz < nops ? l2 + psz : nops, lane);
        }
        if (r.b == WGA_NONE) r.b = nops;
        const u64 nb = chain_walk<false>(rec + r.a, r.b - r.a, (u8*)0, j == 0, r.b == nops, lane, slot, p10, tbuf, st, &weird,
                                         &bad);
        r.nb = weird ? WGA_NONE : nb;
        if (bad != WGA_NONE && lane == 0 && diag) atomicMin((u64*)&diag[i].bad_op_idx, r.a + bad);
        if (j == 0 && !weird && trims) { /* wave-uniform */
          wga_chain_trim tr;
          tr.head_ins = st.seen_m ? (u64)st.head_ins : (u64)st.c_td;
          tr.head_del = st.seen_m ? (u64)st.head_del : (u64)st.c_qd;
          chain_tail_trim(rec, nops, lane, tr.tail_ins, tr.tail_del);
          if (lane == 0) trims[i] = tr;
        }
      }
      if (lane == 0) pc[p] = r;
    } else {
      const wga_chain_piece r = pc[p];
      u8* const text = out + out_off[i];
      if (r.off == WGA_NONE) {
        if (j == 0 && lane == 0) {
          wga_chain_trim tr;
          u64 nbs = 0, bd = WGA_NONE;
          chain_serial(rec, nops, text, tr, nbs, bd);
        }
      } else if (r.a != WGA_NONE) {
        chain_walk<true>(rec + r.a, r.b - r.a, text + r.off, j == 0, r.b == nops, lane, slot, p10, tbuf, st, &weird, &bad);
      }
    }
    WGA_WAVE_SYNC(); /* the next piece reuses the wave's slots and text buffer */
  }
}
/* one thread per long record: the record's bytes, every piece's place in the text */
__global__ __launch_bounds__(256) void k_cigar_chain_piece_scan(u32 n, const u32* __restrict__ ops,
                                                                const u64* __restrict__ op_off,
                                                                const u64* __restrict__ piece_off, wga_chain_piece* pc,
                                                                wga_chain_trim* trims, u64* nbytes, wga_rec_diag* diag) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 p0 = piece_off[i], p1 = piece_off[i + 1];
  if (p0 == p1) return;
  u64 sum = 0;
  bool weird = false;
  for (u64 p = p0; p < p1; p++) {
    const u64 nb = pc[p].nb;
    weird = weird || nb == WGA_NONE;
    pc[p].off = sum;
    sum += nb;
  }
  if (weird) {
    const u64 o0 = op_off[i];
    wga_chain_trim tr;
    u64 bad = WGA_NONE;
    chain_serial(ops + o0, op_off[i + 1] - o0, (u8*)0, tr, sum, bad);
    if (nbytes) trims[i] = tr, diag[i].bad_op_idx = bad;
    for (u64 p = p0; p < p1; p++) pc[p].off = WGA_NONE;
  }
  if (nbytes) nbytes[i] = sum;
}

/* ============================================================================================ */
/* K11: bridges between the run / data-line lists and the packed-op and CIGAR-text forms        */
/*      (maf2chain, chain2paf, chain2maf, maf2paf's cg:Z: text)      */
/* ============================================================================================ */
/* All four share one skeleton: element x (a K3 run, or a chain data line) of record r produces
 * src.size(x, r) output units (packed ops or text bytes); an exclusive scan over the elements
 * gives every element its place inside its record's output, which starts at out_off[r].  One
 * thread per element; its record is found by bisection in the CSR offsets. */
__device__ __forceinline__ u32 csr_find_rec(const u64* __restrict__ off, u32 n, u64 x) {
  u32 lo = 0, hi = n; /* largest r < n with off[r] <= x (records without elements are skipped) */
  while (hi - lo > 1u) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (off[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ u32 split_pieces(u64 len) { /* pieces of at most WGA_OP_MAX_LEN, none for 0 */
  return (u32)((len + (u64)WGA_OP_MAX_LEN - 1ull) / (u64)WGA_OP_MAX_LEN);
}
__device__ __forceinline__ u32* put_split(u32* p, u64 len, u32 code, u32 cont) {
  bool first = true;
  while (len) {
    const u64 piece = len > (u64)WGA_OP_MAX_LEN ? (u64)WGA_OP_MAX_LEN : len;
    *p++ = ((u32)piece << 4) | (first ? code : cont);
    len -= piece;
    first = false;
  }
  return p;
}
__device__ __forceinline__ u8* put_len_op(u8* p, u64 len, u8 op) {
  const u32 nd = dec_digits(len);
  dec_write(p, len, nd);
  p[nd] = op;
  return p + nd + 1u;
}

/* K3 runs (start_column << 3 | class, class 0 '=' 1 I 2 D 3 X) of MAF column pairs */
struct MafRunSrc {
  const u64* runs;
  const u64* run_off;
  const u64* cols;
  __device__ u64 len(u64 x, u32 r) const {
    const u64 start = runs[x] >> 3;
    const u64 end = x + 1 < run_off[r + 1] ? runs[x + 1] >> 3 : cols[r];
    return end - start;
  }
  __device__ u32 cls(u64 x) const { return (u32)(runs[x] & 7ull); }
};
/* -> packed ops: '=' 7, I 1, D 2, X 8; a run of 2^28 columns or more is split like the PAF packer
 * splits a length (continuation codes for I / D), so every consumer of a wga_cigar_batch applies */
struct MafRunOps {
  typedef u32 out_t;
  MafRunSrc s;
  __device__ u64 size(u64 x, u32 r) const { return split_pieces(s.len(x, r)); }
  __device__ void write(u64 x, u32 r, u32* p) const {
    const u32 c = s.cls(x);
    const u32 code = c == 0u ? (u32)WGA_OP_EQ : c == 1u ? (u32)WGA_OP_I : c == 2u ? (u32)WGA_OP_D : (u32)WGA_OP_X;
    const u32 cont = c == 1u ? (u32)WGA_OP_I_CONT : c == 2u ? (u32)WGA_OP_D_CONT : code;
    put_split(p, s.len(x, r), code, cont);
  }
};
/* -> the cg:Z: text of maf2paf, "<len><=|I|D|X>" per run (maf.rs:484-520, cigar.rs:400-401) */
struct MafRunText {
  typedef u8 out_t;
  MafRunSrc s;
  __device__ u64 size(u64 x, u32 r) const { return dec_digits(s.len(x, r)) + 1u; }
  __device__ void write(u64 x, u32 r, u8* p) const {
    const u32 c = s.cls(x);
    put_len_op(p, s.len(x, r), c == 0u ? (u8)'=' : c == 1u ? (u8)'I' : c == 2u ? (u8)'D' : (u8)'X');
  }
};
/* chain data lines, three u64 each: size, 2nd column (bases only in the target: D), 3rd column
 * (bases only in the query: I) — chain.rs:330-348 reads them in this order */
struct ChainLineSrc {
  const u64* lines;
  __device__ u64 size_(u64 x) const { return lines[3 * x]; }
  __device__ u64 del_(u64 x) const { return lines[3 * x + 1]; }
  __device__ u64 ins_(u64 x) const { return lines[3 * x + 2]; }
};
/* -> packed ops in the order parse_chain_to_cigar / parse_chain_to_insert walk a line
 * (cigar.rs:576-606, converter.rs:360-388): M size, I 3rd column, D 2nd column; zero lengths
 * have no effect on rows or counts and are left out */
struct ChainLineOps {
  typedef u32 out_t;
  ChainLineSrc s;
  __device__ u64 size(u64 x, u32) const {
    return (u64)split_pieces(s.size_(x)) + split_pieces(s.ins_(x)) + split_pieces(s.del_(x));
  }
  __device__ void write(u64 x, u32, u32* p) const {
    p = put_split(p, s.size_(x), (u32)WGA_OP_M, (u32)WGA_OP_M);
    p = put_split(p, s.ins_(x), (u32)WGA_OP_I, (u32)WGA_OP_I_CONT);
    put_split(p, s.del_(x), (u32)WGA_OP_D, (u32)WGA_OP_D_CONT);
  }
};
/* -> chain2paf's CIGAR text: "<size>M" always, "<n>I" / "<n>D" when non-zero (cigar.rs:576-606) */
struct ChainLineText {
  typedef u8 out_t;
  ChainLineSrc s;
  __device__ u64 size(u64 x, u32) const {
    const u64 i = s.ins_(x), d = s.del_(x);
    return (u64)dec_digits(s.size_(x)) + 1u + (i ? dec_digits(i) + 1u : 0u) + (d ? dec_digits(d) + 1u : 0u);
  }
  __device__ void write(u64 x, u32, u8* p) const {
    const u64 i = s.ins_(x), d = s.del_(x);
    p = put_len_op(p, s.size_(x), (u8)'M');
    if (i) p = put_len_op(p, i, (u8)'I');
    if (d) put_len_op(p, d, (u8)'D');
  }
};

template <typename F>
struct ScanElem { /* scan functor: output units of element x */
  F f;
  const u64* elem_off;
  u32 n;
  __device__ u64 operator()(u32 x) const { return f.size((u64)x, csr_find_rec(elem_off, n, (u64)x)); }
};
__global__ __launch_bounds__(256) void k_elem_rec_totals(u32 n, const u64* __restrict__ elem_off,
                                                         const u64* __restrict__ esc, u64* __restrict__ cnt) {
  const u32 r = blockIdx.x * 256u + threadIdx.x;
  if (r < n) cnt[r] = esc[elem_off[r + 1]] - esc[elem_off[r]];
}
/* bytes [a, a + total) of an LDS text buffer go to gb + a (gb 16-byte aligned: the buffer mirrors the output's position
 * inside its 16-byte group): whole groups with 16-byte stores, the ragged head and tail (< 16 bytes each) by bytes.
 * `nthr` threads share the work (a wave or a block; the caller synchronises around the call). */
__device__ __forceinline__ void lds_text_flush(const u8* tbuf, u32 a, u32 total, u8* gb, u32 tid, u32 nthr) {
  const u32 end = a + total;
  const u32 g_lo = (a + 15u) >> 4, g_hi = end >> 4; /* whole 16-byte groups [g_lo, g_hi) */
  for (u32 g = g_lo + tid; g < g_hi; g += nthr) *(u32x4_a16*)(gb + 16u * g) = *(const u32x4_a16*)(tbuf + 16u * g);
  const u32 head_end = 16u * g_lo < end ? 16u * g_lo : end;           /* [a, head_end) */
  const u32 tail_beg = 16u * g_hi > head_end ? 16u * g_hi : head_end; /* [tail_beg, end) */
  if (tid < 16u) {
    const u32 x = a + tid;
    if (x < head_end) gb[x] = tbuf[x];
  } else if (tid < 32u) {
    const u32 x = tail_beg + (tid - 16u);
    if (x < end) gb[x] = tbuf[x];
  }
}

/* One thread per element, 256 consecutive elements per block.  The records of the block's first and last element are
 * found once (two wave-wide searches per block); every thread then looks inside that window — one record in nearly every block.
 * A block whose elements belong to ONE record writes one contiguous stretch of that record's output: its threads put
 * their units into an LDS buffer that mirrors the stretch's position inside its 16-byte group, and the stretch goes out
 * in 16-byte stores.  Blocks across a record border, or with more output than the buffer holds, write directly. */
#define WGA_ELEM_STAGE 16384u
template <typename F>
__global__ __launch_bounds__(256) void k_elem_fill(F f, u32 n, u32 ne, const u64* __restrict__ elem_off,
                                                   const u64* __restrict__ esc, typename F::out_t* out,
                                                   const u64* __restrict__ out_off) {
  typedef typename F::out_t out_t;
  __shared__ u32x4_a16 s_buf[(WGA_ELEM_STAGE + 32u) / 16u];
  __shared__ u32 s_r[2];
  const u32 tid = threadIdx.x;
  const u32 x0 = blockIdx.x * 256u, x1 = x0 + 256u < ne ? x0 + 256u : ne;
  /* two waves search, 64 probes a step (three steps for 10^5 records where one thread's bisection takes seventeen) */
  if (tid < 128u) { /* wave-uniform */
    const u32 r = wga_find_rec(elem_off, n, tid < 64u ? (u64)x0 : (u64)(x1 - 1u));
    if ((tid & 63u) == 0u) s_r[tid >> 6] = r;
  }
  __syncthreads();
  const u32 r_lo = WGA_UNI32(s_r[0]), r_hi = WGA_UNI32(s_r[1]);
  const u32 x = x0 + tid;
  const u64 e0 = esc[x0], e1 = esc[x1]; /* units in front of the block, and behind it */
  const bool staged = r_lo == r_hi && (e1 - e0) * sizeof(out_t) <= (u64)WGA_ELEM_STAGE;
  if (staged) { /* block-uniform */
    out_t* const g0 = out + out_off[r_lo] + (e0 - esc[elem_off[r_lo]]);
    const u32 a = (u32)((uintptr_t)g0 & 15u);
    u8* const tbuf = (u8*)s_buf;
    if (x < x1) f.write((u64)x, r_lo, (out_t*)(tbuf + a) + (esc[x] - e0));
    __syncthreads();
    lds_text_flush(tbuf, a, (u32)((e1 - e0) * sizeof(out_t)), (u8*)g0 - a, tid, 256u);
    return;
  }
  if (x >= x1) return;
  u32 lo = r_lo, hi = r_hi + 1u; /* largest r in [r_lo, r_hi] with elem_off[r] <= x */
  while (hi - lo > 1u) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (elem_off[mid] <= (u64)x) lo = mid; else hi = mid;
  }
  f.write((u64)x, lo, out + out_off[lo] + (esc[x] - esc[elem_off[lo]]));
}

/* ============================================================================================ */
/* K12: dotplot base-level segments (emit_baseplotdatas, cigar.rs:815-914) */
/* ============================================================================================ */
/* The reference folds the ops with a `last_m` flag: an I / D longer than the cutoff is a segment
 * of its own and closes the open M segment; an M-like op opens an M segment unless one is open;
 * every other M / small I / small D moves the end of the open segment to the running offsets
 * (small indels with no open segment only advance the offsets; ops outside M = X I D are
 * ignored).  Read as intervals between "breaks" (long indels): an interval holds one M segment
 * iff it has an M-like op; it starts at the offsets of the first such op and ends at the offsets
 * at the end of the interval.  So the opener writes the start fields and the closer — the next
 * break, or the end of the record — writes the end fields, and nothing is serial.
 * One wave per record, 256 ops per step.  Segment = 5 u64: ref_start, ref_end, query_start,
 * query_end (the two swapped for '-' records, cigar.rs:807-812), kind 0 M / 1 I / 2 D.
 * A record with a split (>= 2^28) I / D goes through the serial walk: its pieces count as one op. */
#define WGA_SEG_WORDS 5u
__device__ __forceinline__ void seg_write(u64* s, u64 rs, u64 re, u64 qs, u64 qe, u64 kind, bool neg) {
  s[0] = rs;
  s[1] = re;
  s[2] = neg ? qe : qs;
  s[3] = neg ? qs : qe;
  s[4] = kind;
}
__device__ __forceinline__ u64 dotplot_serial(const u32* rec, u64 nops, u64 cutoff, u64 r, u64 q, bool neg,
                                              u64* segs) {
  u64 ns = 0;
  bool last_m = false;
  for (u64 k = 0; k < nops;) {
    const u32 code = rec[k] & 15u;
    u64 len = rec[k] >> 4;
    u64 k2 = k + 1;
    if (code == WGA_OP_I || code == WGA_OP_D) /* pieces of one split length */
      while (k2 < nops && (rec[k2] & 15u) == (code == WGA_OP_I ? (u32)WGA_OP_I_CONT : (u32)WGA_OP_D_CONT)) len += rec[k2++] >> 4;
    const bool isi = code == WGA_OP_I || code == WGA_OP_I_CONT, isd = code == WGA_OP_D || code == WGA_OP_D_CONT;
    if (code == WGA_OP_M || code == WGA_OP_EQ || code == WGA_OP_X) {
      if (!last_m) {
        if (segs) seg_write(segs + ns * WGA_SEG_WORDS, r, r + len, q, q + len, 0, neg);
        ns++;
      } else if (segs) {
        u64* s = segs + (ns - 1) * WGA_SEG_WORDS;
        s[1] = r + len;
        s[neg ? 2 : 3] = q + len;
      }
      r += len;
      q += len;
      last_m = true;
    } else if (isi || isd) {
      const u64 re = isd ? r + len : r, qe = isi ? q + len : q;
      if (len > cutoff) {
        if (segs) seg_write(segs + ns * WGA_SEG_WORDS, r, re, q, qe, isi ? 1 : 2, neg);
        ns++;
        last_m = false;
      } else if (last_m && segs) {
        u64* s = segs + (ns - 1) * WGA_SEG_WORDS;
        if (isd) s[1] = re; else s[neg ? 2 : 3] = qe;
      }
      r = re;
      q = qe;
    }
    k = k2;
  }
  return ns;
}

/* The walk of ops [a, b) of one record by one wave (a a multiple of 256): offsets r / q, the number of segments so far and the
 * open / closed state in front of the range come in through `st` and leave through it; `first_ev` = the first event of the
 * range (0 none, 1 a break, 2 an M-like op) tells the caller whether a range that follows an open M segment starts a new
 * one.  Returns false when the range holds a continuation piece of a split indel (the record then takes the serial walk). */
struct DotState {
  u64 r, q, nseg;
  u32 state; /* 0 / 1: no open M segment (start, or a break was the last event), 2: open */
};
template <bool FILL>
__device__ __forceinline__ bool dotplot_walk(const u32* __restrict__ rec, u64 a, u64 b, u64 cutoff, bool neg, DotState& ds,
                                             u64* out, u32 lane, u32* first_ev) {
  u64 r_base = ds.r, q_base = ds.q, nseg = ds.nseg;
  u32 carry_state = ds.state, first_seen = 0u;
  u32 wnext[4]; /* the next step's ops travel behind the work on this step's */
#pragma unroll
  for (int e = 0; e < 4; e++) wnext[e] = a + (u64)lane * 4u + (u64)e < b ? rec[a + (u64)lane * 4u + e] : 0xFu;
  for (u64 k0 = a; k0 < b; k0 += 256) {
    const u64 kb = k0 + (u64)lane * 4u;
    u32 len[4], radv[4], qadv[4];
    bool ml[4], brk[4], isi[4];
    bool cont = false;
    u32 sr = 0, sq = 0, last_ev = 0, lane_first = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const u32 w = wnext[e];
      wnext[e] = kb + 256u + (u64)e < b ? rec[kb + 256u + e] : 0xFu;
      const u32 code = w & 15u;
      len[e] = w >> 4;
      ml[e] = code == WGA_OP_M || code == WGA_OP_EQ || code == WGA_OP_X;
      isi[e] = code == WGA_OP_I;
      const bool isd = code == WGA_OP_D;
      cont |= code == WGA_OP_I_CONT || code == WGA_OP_D_CONT;
      brk[e] = (isi[e] || isd) && (u64)len[e] > cutoff;
      radv[e] = (ml[e] || isd) ? len[e] : 0u;
      qadv[e] = (ml[e] || isi[e]) ? len[e] : 0u;
      sr += radv[e];
      sq += qadv[e];
      const u32 ev = ml[e] ? 2u : brk[e] ? 1u : 0u;
      last_ev = ev ? ev : last_ev;
      lane_first = lane_first ? lane_first : ev;
    }
    if (__ballot(cont)) return false; /* wave-uniform */
    /* offsets in front of this lane's ops: exact wave scans of the lane sums (< 2^30) on 16-bit halves */
    const u32 rl = wave_incl_scan_u32(sr & 0xFFFFu), rh = wave_incl_scan_u32(sr >> 16);
    const u32 ql = wave_incl_scan_u32(sq & 0xFFFFu), qh = wave_incl_scan_u32(sq >> 16);
    u64 r = r_base + (((u64)rh << 16) + (u64)rl) - (u64)sr;
    u64 q = q_base + (((u64)qh << 16) + (u64)ql) - (u64)sq;
    /* open / closed in front of this lane = the last event of the nearest earlier lane that has one */
    const u64 all_ev = __ballot(last_ev != 0u);
    const u64 evm = all_ev & ((1ull << lane) - 1ull);
    const int src = evm ? 63 - (int)__builtin_clzll(evm) : 0;
    const u32 got = (u32)__shfl((int)last_ev, src);
    u32 state = evm ? got : carry_state;
    if (!first_seen && all_ev) first_seen = (u32)__shfl((int)lane_first, (int)__builtin_ctzll(all_ev));
    /* segments this lane raises, then their ranks */
    u32 st = state, cnt = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      cnt += (brk[e] || (ml[e] && st != 2u)) ? 1u : 0u;
      st = ml[e] ? 2u : brk[e] ? 1u : st;
    }
    const u32 cinc = wave_incl_scan_u32(cnt);
    if (FILL) {
      u64 idx = nseg + (u64)(cinc - cnt);
      st = state;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (brk[e]) {
          if (st == 2u) { /* closes the open M segment */
            u64* s = out + (idx - 1) * WGA_SEG_WORDS;
            s[1] = r;
            s[neg ? 2 : 3] = q;
          }
          seg_write(out + idx * WGA_SEG_WORDS, r, r + radv[e], q, q + qadv[e], isi[e] ? 1 : 2, neg);
          idx++;
        } else if (ml[e] && st != 2u) { /* opens one: the end fields come from its closer */
          u64* s = out + idx * WGA_SEG_WORDS;
          s[0] = r;
          s[neg ? 3 : 2] = q;
          s[4] = 0;
          idx++;
        }
        st = ml[e] ? 2u : brk[e] ? 1u : st;
        r += radv[e];
        q += qadv[e];
      }
    }
    nseg += (u64)wave_last_u32(cinc);
    r_base += ((u64)wave_last_u32(rh) << 16) + (u64)wave_last_u32(rl);
    q_base += ((u64)wave_last_u32(qh) << 16) + (u64)wave_last_u32(ql);
    if (all_ev) carry_state = (u32)__shfl((int)last_ev, 63 - (int)__builtin_clzll(all_ev));
  }
  ds.r = r_base;
  ds.q = q_base;
  ds.nseg = nseg;
  ds.state = carry_state;
  *first_ev = first_seen;
  return true;
}

/* one wave per record; records beyond `long_ops` ops are left to the piece kernels below (long_ops = 0: none are) */
template <bool FILL>
__global__ __launch_bounds__(256) void k_dotplot_segments(u32 n, const u32* __restrict__ ops,
                                                          const u64* __restrict__ op_off,
                                                          const u8* __restrict__ strand_neg, u64 cutoff,
                                                          const u64* __restrict__ t_start,
                                                          const u64* __restrict__ q_start, u64* seg_cnt,
                                                          u64* segs, const u64* seg_off, u64 long_ops) {
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 i = (u64)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
  if (long_ops && nops > long_ops) return;
  const u32* rec = ops + o0;
  const bool neg = strand_neg[i] != 0;
  u64* const out = FILL ? segs + seg_off[i] * WGA_SEG_WORDS : (u64*)0;
  DotState ds;
  ds.r = t_start[i], ds.q = q_start[i], ds.nseg = 0, ds.state = 0u;
  u32 first_ev;
  if (!dotplot_walk<FILL>(rec, 0, nops, cutoff, neg, ds, out, lane, &first_ev)) { /* wave-uniform */
    if (lane == 0) {
      const u64 ns = dotplot_serial(rec, nops, cutoff, t_start[i], q_start[i], neg, out);
      if (!FILL) seg_cnt[i] = ns;
    }
    return;
  }
  if (lane == 0) {
    if (FILL && ds.state == 2u) { /* the end of the record closes the open M segment */
      u64* s = out + (ds.nseg - 1) * WGA_SEG_WORDS;
      s[1] = ds.r;
      s[neg ? 2 : 3] = ds.q;
    }
    if (!FILL) seg_cnt[i] = ds.nseg;
  }
}

/* ---- long records in pieces (see the piece kernels of K7): a first walk leaves each piece's offset sums, its segment count
 *      as if nothing were open in front of it, its first event and the state behind it; one thread per record turns that into
 *      each piece's start (offsets, segments so far, open / closed); the second walk writes the segments and the record's
 *      last piece closes the segment that is still open. ----------------------------------------------------------------- */
struct wga_dot_piece {
  u64 r, q, nseg; /* MODE 0: sums / count (start closed); after the record scan: the piece's start */
  u32 state;      /* MODE 0: state behind the piece, 0 when it has no event; after the scan: state in front of it */
  u32 first_ev;   /* MODE 0: 0 none, 1 break, 2 M-like, 3: a continuation piece of a split indel (serial walk); after the
                     scan: 3 = the record takes the serial walk */
};
template <int MODE>
__global__ __launch_bounds__(256) void k_dotplot_pieces(u32 n, const u32* __restrict__ ops, const u64* __restrict__ op_off,
                                                        const u8* __restrict__ strand_neg, u64 cutoff,
                                                        const u64* __restrict__ t_start, const u64* __restrict__ q_start,
                                                        const u64* __restrict__ piece_off, const u32* __restrict__ piece_rec, wga_dot_piece* pc,
                                                        u64* segs, const u64* seg_off) {
  const u32 lane = threadIdx.x & 63u;
  const u64 n_pieces = piece_off[n];
  const u64 n_waves = (u64)gridDim.x * 4u;
  for (u64 p = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x); p < n_pieces; p += n_waves) {
    const u32 i = WGA_UNI32(piece_rec[p]);
    const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
    const u64 psz = piece_span(nops, piece_off[i + 1] - piece_off[i]);
    const u64 a0 = (p - piece_off[i]) * psz, a = a0 < nops ? a0 : nops, b = a + psz < nops ? a + psz : nops;
    const bool neg = strand_neg[i] != 0;
    DotState ds;
    u32 first_ev = 0u;
    if (MODE == 0) {
      ds.r = ds.q = ds.nseg = 0;
      ds.state = 0u;
      const bool ok = dotplot_walk<false>(ops + o0, a, b, cutoff, neg, ds, (u64*)0, lane, &first_ev);
      if (lane == 0) {
        wga_dot_piece r;
        r.r = ds.r, r.q = ds.q, r.nseg = ds.nseg;
        r.state = first_ev ? ds.state : 0u;
        r.first_ev = ok ? first_ev : 3u;
        pc[p] = r;
      }
    } else {
      const wga_dot_piece r = pc[p];
      u64* const out = segs + seg_off[i] * WGA_SEG_WORDS;
      if (r.first_ev == 3u) { /* a split indel somewhere in the record: the serial walk, by the first piece's first lane */
        if (a == 0 && lane == 0) dotplot_serial(ops + o0, nops, cutoff, t_start[i], q_start[i], neg, out);
        continue;
      }
      ds.r = r.r, ds.q = r.q, ds.nseg = r.nseg, ds.state = r.state;
      dotplot_walk<true>(ops + o0, a, b, cutoff, neg, ds, out, lane, &first_ev);
      if (b == nops && lane == 0 && ds.state == 2u) { /* the end of the record closes the open M segment */
        u64* s = out + (ds.nseg - 1) * WGA_SEG_WORDS;
        s[1] = ds.r;
        s[neg ? 2 : 3] = ds.q;
      }
    }
  }
}
/* one thread per long record: piece sums -> piece starts; the record's segment count */
__global__ __launch_bounds__(256) void k_dotplot_piece_scan(u32 n, const u32* __restrict__ ops, const u64* __restrict__ op_off,
                                                            const u8* __restrict__ strand_neg, u64 cutoff,
                                                            const u64* __restrict__ t_start, const u64* __restrict__ q_start,
                                                            const u64* __restrict__ piece_off, wga_dot_piece* pc, u64* seg_cnt) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 p0 = piece_off[i], p1 = piece_off[i + 1];
  if (p0 == p1) return;
  bool weird = false;
  for (u64 p = p0; p < p1; p++) weird |= pc[p].first_ev == 3u;
  if (weird) {
    for (u64 p = p0; p < p1; p++) pc[p].first_ev = 3u;
    if (seg_cnt)
      seg_cnt[i] = dotplot_serial(ops + op_off[i], op_off[i + 1] - op_off[i], cutoff, t_start[i], q_start[i], strand_neg[i] != 0, (u64*)0);
    return;
  }
  u64 r = t_start[i], q = q_start[i], ns = 0;
  u32 state = 0u;
  for (u64 p = p0; p < p1; p++) {
    const wga_dot_piece x = pc[p];
    wga_dot_piece s;
    s.r = r, s.q = q, s.nseg = ns, s.state = state, s.first_ev = x.first_ev;
    pc[p] = s;
    r += x.r;
    q += x.q;
    /* counted as if nothing were open: an M-like first event continues the segment that is */
    ns += x.nseg - ((state == 2u && x.first_ev == 2u) ? 1u : 0u);
    state = x.first_ev ? x.state : state;
  }
  if (seg_cnt) seg_cnt[i] = ns;
}

/* ============================================================================================ */
/* K13: PAF field splitter (the csv / serde floor of paf.rs:24-30,50-78)   */
/* ============================================================================================ */
/* The file text goes to the device once; the bytes that can end a field or a record — tab, newline,
 * and the two that need the csv crate's full state machine, '"' and '\r' — are listed in order
 * (count, scan, fill over 4 KB blocks), with the newlines' ranks in that list.  Then one thread per
 * line reads its dozen-odd delimiters: the 12 fixed fields (u64::from_str, Strand), the two name
 * spans and the span of the cg:Z: text, which the tokeniser reads in place.  Anything outside the
 * plain case — a quote or CR on the line, fewer than 12 fields, a bad integer or strand, a cs:Z:
 * tag standing in for cg:Z: — marks the line WGA_PAF_FALLBACK and the caller re-reads the file with
 * its csv-semantics parser (same records, or the reference's exact error).  4 B read twice per byte
 * of text; the per-line pass is negligible. */
#define WGA_PAF_OK 0
#define WGA_PAF_SKIP 1     /* blank line or '#' comment (csv reader: skipped) */
#define WGA_PAF_FALLBACK 2
struct wga_paf_line_dev {
  u64 num[9]; /* query_length, query_start, query_end, target_length, target_start, target_end, matches, block_length, mapq */
  u64 qname_off, tname_off, cg_beg, cg_end; /* byte offsets in the text; cg_beg == WGA_NONE: no cg:Z: tag */
  u32 qname_len, tname_len, n_fields;
  u8 strand_neg, status, pad[2];
};

/* MODE 0 (PAF): tab, newline, '"', CR.  MODE 1 (MAF): newline, the ASCII white space of
 * split_whitespace (9-13, 32) and every byte >= 0x80 (Unicode white space: left to the host). */
template <int MODE>
__device__ __forceinline__ u32 paf_delim_masks(const u32 w[4], u32* nl_mask) {
  u32 dm = 0, nm = 0;
#pragma unroll
  for (int d = 0; d < 4; d++) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const u32 ch = (w[d] >> (8 * b)) & 0xFFu;
      const u32 bit = 1u << (4 * d + b);
      nm |= ch == 0x0Au ? bit : 0u;
      if (MODE == 0)
        dm |= (ch == 0x0Au || ch == 0x09u || ch == 0x22u || ch == 0x0Du) ? bit : 0u;
      else
        dm |= (ch - 9u <= 4u || ch == 0x20u || ch >= 0x80u) ? bit : 0u;
    }
  }
  *nl_mask = nm;
  return dm;
}

template <bool FILL, int MODE>
__global__ __launch_bounds__(256) void k_paf_delims(const u8* __restrict__ text, u64 n_bytes, u64* blk,
                                                    const u64* blk_off, u64* delims, u64* nl_idx) {
  __shared__ u64 s_w[5];
  const u64 c = ((u64)blockIdx.x * 256u + threadIdx.x) * 16u;
  u32 w[4] = {0, 0, 0, 0};
  if (c + 16u <= n_bytes) {
    const u32x4_a1 a = *(const u32x4_a1*)(text + c);
    w[0] = a[0], w[1] = a[1], w[2] = a[2], w[3] = a[3];
  } else if (c < n_bytes) {
    for (u32 j = 0; j < (u32)(n_bytes - c); j++) w[j >> 2] |= (u32)text[c + j] << (8u * (j & 3u));
  }
  u32 nm;
  u32 dm = paf_delim_masks<MODE>(w, &nm);
  const u64 cnt = (u64)__builtin_popcount(dm) | ((u64)__builtin_popcount(nm) << 32);
  u64 tot;
  const u64 ex = block_excl_scan_u64(cnt, s_w, &tot);
  if (!FILL) {
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
    return;
  }
  const u64 base = blk_off[blockIdx.x] + ex; /* both halves advance independently: totals < 2^32 */
  u64 di = base & 0xFFFFFFFFull, ni = base >> 32;
  while (dm) {
    const u32 j = (u32)__builtin_ctz(dm);
    dm &= dm - 1u;
    delims[di] = c + j;
    if ((nm >> j) & 1u) nl_idx[ni++] = di;
    di++;
  }
}

/* u64::from_str on text[a, b): optional '+', >= 1 digit, no overflow */
__device__ __forceinline__ bool paf_parse_u64(const u8* __restrict__ text, u64 a, u64 b, u64* out) {
  if (a < b && text[a] == (u8)'+') a++;
  if (a >= b || b - a > 20u) return false;
  u64 v = 0;
  for (u64 k = a; k < b; k++) {
    const u32 d = (u32)text[k] - 0x30u;
    if (d > 9u) return false;
    if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10ull) return false;
    v = v * 10ull + d;
  }
  *out = v;
  return true;
}

__global__ __launch_bounds__(256) void k_paf_fields(const u8* __restrict__ text, u64 n_bytes, u64 n_lines,
                                                    u64 n_newlines, u64 n_delims,
                                                    const u64* __restrict__ delims,
                                                    const u64* __restrict__ nl_idx, wga_paf_line_dev* lines) {
  const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
  if (j >= n_lines) return;
  /* line j = bytes [s, e), its inner delimiters = delims[d0, d1) */
  const u64 d0 = j ? nl_idx[j - 1] + 1 : 0;
  const u64 s = j ? delims[nl_idx[j - 1]] + 1 : 0;
  const u64 d1 = j < n_newlines ? nl_idx[j] : n_delims;
  const u64 e = j < n_newlines ? delims[nl_idx[j]] : n_bytes;
  wga_paf_line_dev L;
  for (int k = 0; k < 9; k++) L.num[k] = 0;
  L.qname_off = L.tname_off = 0;
  L.cg_beg = L.cg_end = WGA_NONE;
  L.qname_len = L.tname_len = L.n_fields = 0;
  L.strand_neg = 0;
  L.pad[0] = L.pad[1] = 0;
  u8 status = WGA_PAF_OK;
  if (s == e || text[s] == (u8)'#') {
    status = WGA_PAF_SKIP;
    for (u64 d = d0; d < d1 && status == WGA_PAF_SKIP; d++) /* a '\r' inside would end the comment for csv */
      if (text[delims[d]] == 0x0Du) status = WGA_PAF_FALLBACK;
  } else {
    u64 fs = s; /* start of the current field */
    u32 nf = 0;
    bool seen_cg = false, seen_cs = false;
    for (u64 d = d0; d <= d1 && status == WGA_PAF_OK; d++) {
      u64 fe = e;
      if (d < d1) {
        fe = delims[d];
        if (text[fe] != 0x09u) { /* a quote or CR: the csv state machine decides */
          status = WGA_PAF_FALLBACK;
          break;
        }
      }
      bool ok = true;
      switch (nf) {
        case 0: L.qname_off = fs; L.qname_len = (u32)(fe - fs); ok = fe - fs < 0xFFFFFFFFull; break;
        case 1: ok = paf_parse_u64(text, fs, fe, &L.num[0]); break;
        case 2: ok = paf_parse_u64(text, fs, fe, &L.num[1]); break;
        case 3: ok = paf_parse_u64(text, fs, fe, &L.num[2]); break;
        case 4:
          ok = fe - fs == 1u && (text[fs] == (u8)'+' || text[fs] == (u8)'-');
          L.strand_neg = ok && text[fs] == (u8)'-' ? 1 : 0;
          break;
        case 5: L.tname_off = fs; L.tname_len = (u32)(fe - fs); ok = fe - fs < 0xFFFFFFFFull; break;
        case 6: ok = paf_parse_u64(text, fs, fe, &L.num[3]); break;
        case 7: ok = paf_parse_u64(text, fs, fe, &L.num[4]); break;
        case 8: ok = paf_parse_u64(text, fs, fe, &L.num[5]); break;
        case 9: ok = paf_parse_u64(text, fs, fe, &L.num[6]); break;
        case 10: ok = paf_parse_u64(text, fs, fe, &L.num[7]); break;
        case 11: ok = paf_parse_u64(text, fs, fe, &L.num[8]); break;
        default: /* tags: the first cg:Z: wins (paf.rs:126-130), a cs:Z: is only used without one */
          if (fe - fs >= 5u && text[fs] == (u8)'c' && text[fs + 2] == (u8)':' && text[fs + 3] == (u8)'Z' &&
              text[fs + 4] == (u8)':') {
            if (text[fs + 1] == (u8)'g' && !seen_cg) {
              seen_cg = true;
              L.cg_beg = fs + 5u;
              L.cg_end = fe;
            } else if (text[fs + 1] == (u8)'s') {
              seen_cs = true;
            }
          }
          break;
      }
      if (!ok) status = WGA_PAF_FALLBACK;
      nf++;
      fs = fe + 1u;
    }
    L.n_fields = nf;
    if (status == WGA_PAF_OK && (nf < 12u || (!seen_cg && seen_cs))) status = WGA_PAF_FALLBACK;
  }
  L.status = status;
  lines[j] = L;
}

/* ============================================================================================ */
/* K14: MAF line splitter (the reader of maf.rs:25-36,138-211,371-421 for plain files)           */
/* ============================================================================================ */
/* Same two lists as K13, with white space as the field delimiter.  One thread per line: a line that
 * starts with 's' (and is not the file's first line, which is always the header) is an s-line:
 * seven white-space separated tokens — mode, name, start, size, strand, srcSize, text — of which
 * the name and the text stay where they are (spans).  The K3 / K4 walks then read the rows straight
 * out of the uploaded file.  Blocks (maximal runs of s-lines) are put together by the caller. */
#define WGA_MAF_SLINE 0
#define WGA_MAF_OTHER 1    /* header, or a line that does not start with 's': ends a block */
#define WGA_MAF_FALLBACK 2 /* not seven tokens, a bad number or strand, a non-ASCII byte in front of the text */
struct wga_maf_line_dev {
  u64 num[3]; /* start, align_size, size (maf.rs:65-73) */
  u64 name_off, seq_off, seq_len;
  u32 name_len;
  u8 strand_neg, status, pad[2];
};

__global__ __launch_bounds__(256) void k_maf_lines(const u8* __restrict__ text, u64 n_bytes, u64 n_lines,
                                                   u64 n_newlines, u64 n_delims,
                                                   const u64* __restrict__ delims,
                                                   const u64* __restrict__ nl_idx, wga_maf_line_dev* lines) {
  const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
  if (j >= n_lines) return;
  const u64 d0 = j ? nl_idx[j - 1] + 1 : 0;
  const u64 s = j ? delims[nl_idx[j - 1]] + 1 : 0;
  const u64 d1 = j < n_newlines ? nl_idx[j] : n_delims;
  const u64 e = j < n_newlines ? delims[nl_idx[j]] : n_bytes;
  wga_maf_line_dev L;
  L.num[0] = L.num[1] = L.num[2] = 0;
  L.name_off = L.seq_off = L.seq_len = 0;
  L.name_len = 0;
  L.strand_neg = 0;
  L.pad[0] = L.pad[1] = 0;
  u8 status = WGA_MAF_OTHER;
  if (j > 0 && s < e && text[s] == (u8)'s') {
    status = WGA_MAF_SLINE;
    u64 prev = s;
    u32 nt = 0;
    for (u64 d = d0; d <= d1 && status == WGA_MAF_SLINE; d++) {
      const u64 p = d < d1 ? delims[d] : e;
      if (d < d1 && text[p] >= 0x80u) status = WGA_MAF_FALLBACK;
      if (p > prev) { /* a token */
        bool ok = true;
        switch (nt) {
          case 0: break; /* mode: its first char, not looked at again */
          case 1: L.name_off = prev; L.name_len = (u32)(p - prev); ok = p - prev < 0xFFFFFFFFull; break;
          case 2: ok = paf_parse_u64(text, prev, p, &L.num[0]); break;
          case 3: ok = paf_parse_u64(text, prev, p, &L.num[1]); break;
          case 4:
            ok = p - prev == 1u && (text[prev] == (u8)'+' || text[prev] == (u8)'-');
            L.strand_neg = ok && text[prev] == (u8)'-' ? 1 : 0;
            break;
          case 5: ok = paf_parse_u64(text, prev, p, &L.num[2]); break;
          case 6: L.seq_off = prev; L.seq_len = p - prev; break;
          default: ok = false; break; /* SurplusField */
        }
        if (!ok) status = WGA_MAF_FALLBACK;
        nt++;
      }
      prev = p + 1;
    }
    if (status == WGA_MAF_SLINE && nt != 7u) status = WGA_MAF_FALLBACK;
  }
  L.status = status;
  lines[j] = L;
}

/* ============================================================================================ */
/* stat totals: the sum of all records' counters (what `stat` aggregates per pair, stat.rs:181-223, */
/* for one pair; the 88 bytes a multi-GPU run all-reduces)                                         */
/* ============================================================================================ */
/* ============================================================================================ */
/* K15: FASTA text in HBM -> line-stripped sequence pool + contig table   */
/* ============================================================================================ */
/* What the drivers fetch through htslib's faidx (converter.rs:183-184,219-225, paf.rs:221-237, pseudomaf.rs:214-237) is
 * a byte range of a contig's sequence with the line ends taken out.  The file is uploaded as it is; a header line starts
 * with '>' at a line start and runs to its '\n'; every other byte behind the first header that is not a '\n' (nor the '\r'
 * in front of one, nor a '\r' closing the file) is a base of the pool, case preserved.  Three passes over 4 KB blocks:
 * header starts (count, scan, fill — in order), one thread per header for its line end, then the bases (count, scan,
 * compact through LDS, coalesced stores).  pool_off of a contig = the output index at the byte behind its header line. */
struct wga_fa_contig_dev {
  u64 hdr_start, hdr_end, pool_off, len; /* hdr_end = offset of the header line's '\n' (n_bytes if the file ends first) */
};

template <bool FILL>
__global__ __launch_bounds__(256) void k_fa_headers(const u8* __restrict__ text, u64 n_bytes, u64* blk, const u64* blk_off,
                                                    wga_fa_contig_dev* contigs) {
  __shared__ u64 s_w[5];
  const u64 c = ((u64)blockIdx.x * 256u + threadIdx.x) * 16u;
  u32 hm = 0;
  if (c < n_bytes) {
    u8 prev = c == 0 ? (u8)'\n' : text[c - 1];
    const u32 m = n_bytes - c < 16u ? (u32)(n_bytes - c) : 16u;
    for (u32 j = 0; j < m; j++) {
      const u8 ch = text[c + j];
      hm |= (ch == (u8)'>' && prev == (u8)'\n') ? 1u << j : 0u;
      prev = ch;
    }
  }
  u64 tot;
  const u64 ex = block_excl_scan_u64((u64)__builtin_popcount(hm), s_w, &tot);
  if (!FILL) {
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
    return;
  }
  u64 k = blk_off[blockIdx.x] + ex;
  while (hm) {
    const u32 j = (u32)__builtin_ctz(hm);
    hm &= hm - 1u;
    contigs[k++].hdr_start = c + j;
  }
}

__global__ __launch_bounds__(256) void k_fa_header_ends(const u8* __restrict__ text, u64 n_bytes, u64 nh,
                                                        wga_fa_contig_dev* contigs) {
  const u64 k = (u64)blockIdx.x * 256u + threadIdx.x;
  if (k >= nh) return;
  u64 p = contigs[k].hdr_start;
  while (p < n_bytes && text[p] != (u8)'\n') p++;
  contigs[k].hdr_end = p;
}

/* last header with hdr_start <= x, or -1 */
__device__ __forceinline__ i64 fa_find_header(const wga_fa_contig_dev* contigs, u64 nh, u64 x) {
  i64 lo = -1, hi = (i64)nh;
  while (hi - lo > 1) {
    const i64 mid = lo + ((hi - lo) >> 1);
    if (contigs[mid].hdr_start <= x)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_fa_bases(const u8* __restrict__ text, u64 n_bytes, u64 nh,
                                                  wga_fa_contig_dev* contigs, u64* blk, const u64* blk_off, u8* pool) {
  __shared__ u64 s_w[5];
  __shared__ i64 s_k0;
  __shared__ u8 s_out[4096];
  const u64 b0 = (u64)blockIdx.x * 4096u;
  if (threadIdx.x == 0) s_k0 = b0 ? fa_find_header(contigs, nh, b0 - 1u) : -1; /* last header that starts BEFORE the block */
  __syncthreads();
  const u64 c = b0 + (u64)threadIdx.x * 16u;
  i64 k = s_k0;
  u32 keep = 0, m = 0;
  u8 bytes[16];
  if (c < n_bytes) {
    m = n_bytes - c < 16u ? (u32)(n_bytes - c) : 16u;
    while (k + 1 < (i64)nh && contigs[k + 1].hdr_start < c) k++;
  }
  const i64 k_first = k; /* last header that starts before this thread's first byte */
  {
    i64 kk = k_first;
    u64 he = kk >= 0 ? contigs[kk].hdr_end : 0;
    for (u32 j = 0; j < m; j++) {
      const u64 p = c + j;
      if (kk + 1 < (i64)nh && contigs[kk + 1].hdr_start == p) {
        kk++;
        he = contigs[kk].hdr_end;
      }
      const u8 ch = text[p];
      bytes[j] = ch;
      const u8 nx = p + 1u < n_bytes ? text[p + 1u] : (u8)'\n';
      const bool eol = ch == (u8)'\n' || (ch == (u8)'\r' && nx == (u8)'\n');
      keep |= (kk >= 0 && p > he && !eol) ? 1u << j : 0u;
    }
  }
  u64 tot;
  const u64 ex = block_excl_scan_u64((u64)__builtin_popcount(keep), s_w, &tot);
  if (!FILL) {
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
    return;
  }
  const u64 base = blk_off[blockIdx.x];
  { /* the byte right behind a header line fixes that contig's pool_off (checked BEFORE a header that starts there) */
    i64 kk = k_first;
    u64 he = kk >= 0 ? contigs[kk].hdr_end : 0;
    for (u32 j = 0; j < m; j++) {
      const u64 p = c + j;
      if (kk >= 0 && p == he + 1u) contigs[kk].pool_off = base + ex + (u64)__builtin_popcount(keep & ((1u << j) - 1u));
      if (kk + 1 < (i64)nh && contigs[kk + 1].hdr_start == p) {
        kk++;
        he = contigs[kk].hdr_end;
      }
    }
  }
  u32 o = (u32)ex;
  for (u32 j = 0; j < m; j++)
    if ((keep >> j) & 1u) s_out[o++] = bytes[j];
  __syncthreads();
  for (u32 i = threadIdx.x; i < (u32)tot; i += 256u) pool[base + i] = s_out[i];
}

/* contigs whose header line ends the file (or is followed directly by the next header) were not visited by a byte
 * "right behind the header line" inside a sequence region only if the file ends there: give them pool_off = total;
 * then len = next pool_off - pool_off */
__global__ __launch_bounds__(256) void k_fa_finish(u64 n_bytes, u64 nh, u64 total, wga_fa_contig_dev* contigs) {
  const u64 k = (u64)blockIdx.x * 256u + threadIdx.x;
  if (k >= nh) return;
  if (contigs[k].hdr_end + 1u >= n_bytes) contigs[k].pool_off = total;
}
__global__ __launch_bounds__(256) void k_fa_lengths(u64 nh, u64 total, wga_fa_contig_dev* contigs) {
  const u64 k = (u64)blockIdx.x * 256u + threadIdx.x;
  if (k >= nh) return;
  const u64 next = k + 1u < nh ? contigs[k + 1u].pool_off : total;
  contigs[k].len = next - contigs[k].pool_off;
}

/* grid-stride over the n x 11 u64 matrix read as a flat array: thread t always meets field t % 11 when the
 * stride is a multiple of 11; wave sums by shuffles, then one atomic per field and wave */
__global__ __launch_bounds__(256) void k_counts_total(u32 n, const u64* __restrict__ counts, u64* totals) {
  const u64 total = (u64)n * 11ull;
  const u64 stride = (u64)gridDim.x * 253ull; /* 253 = 23 x 11 threads of each block work */
  u64 acc = 0;
  if (threadIdx.x < 253u)
    for (u64 x = (u64)blockIdx.x * 253ull + threadIdx.x; x < total; x += stride) acc += counts[x];
  __shared__ u64 s_acc[256];
  s_acc[threadIdx.x] = threadIdx.x < 253u ? acc : 0ull;
  __syncthreads();
  if (threadIdx.x < 11u) { /* field f = threadIdx.x: threads f, f + 11, ... */
    u64 sum = 0;
    for (u32 k = threadIdx.x; k < 253u; k += 11u) sum += s_acc[k];
    if (sum) atomicAdd(totals + threadIdx.x, sum);
  }
}

#endif /* WGA_KERNELS2_H */
