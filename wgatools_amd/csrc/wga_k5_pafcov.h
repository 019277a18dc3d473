/*
 * wga_k5_pafcov.h — K5: pafcov — difference-array coverage marks (update_cov_vec, cigar.rs:710-741) and the marks -> counts replay.
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K5_PAFCOV_H
#define WGA_K5_PAFCOV_H

#include <type_traits>

#include "wga_kernels.h"

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
#ifndef WGA_COV_WIN_SHIFT
#define WGA_COV_WIN_SHIFT 14u /* 64 KB of counters per block, two blocks per CU: pieces are as long as windows let them be (round 6: 13 -> 14) */
#endif
#define WGA_COV_WIN (1u << WGA_COV_WIN_SHIFT)
/* K5's own tile: WGA_COV_TILE ops per wave, WGA_COV_LO consecutive ones per lane.  1 024 (16 per lane) as everywhere else;
 * 2 048 (-DWGA_COV_TILE_SHIFT=11u: 32 per lane, 128 VGPRs) was measured at configs[3]'s size — the list pass 31.8 ms against
 * 28.0, the replay 34.0 against 35.0 (fewer, longer pieces), 74.8 against 74.2 ms in all (profiles/r05_k5_stated_run5.log). */
#ifndef WGA_COV_TILE_SHIFT
#define WGA_COV_TILE_SHIFT 10u
#endif
#define WGA_COV_TILE (1u << WGA_COV_TILE_SHIFT)
#define WGA_COV_LO (WGA_COV_TILE / 64u)       /* ops per lane */
#define WGA_COV_LO_SHIFT (WGA_COV_TILE_SHIFT - 6u)

struct __attribute__((aligned(16))) wga_cov_piece {
  u32 g;       /* tile */
  u32 ab;      /* first op | end op << 16, tile-relative (<= WGA_COV_TILE): the ops of the record segment whose marks can lie in
                  the window the piece is listed under (whole lanes of WGA_COV_LO ops) */
  u64 pos0;    /* coverage index (cov_off + target position) in front of op `first` */
  u64 limit;   /* coverage index one past the target's last counter */
  u32 wi;      /* window the piece is listed under */
  u32 pad;     /* WGA_COV_NARROW: the segment advances less than 2^27 bases inside its tile */
};
#define WGA_COV_NARROW 1u
#define WGA_COV_NARROW_BITS 27u /* the replay walks such a piece in 32-bit BYTE offsets into its window's counters: 4 x (2^27 in front
                                   + the window + 2^27 + an op's 2^27) stays below 2^31 */
/* What the replay reads per piece, 16 bytes, made from the list pass's piece where it is taken to its window (k_cov_place_*):
 * everything that is the same for all lanes and steps is worked out there, once, by a thread that waits for an atomic anyway,
 * instead of by every replaying wave's scalar unit (round 6: the replay is bound by instruction issue — 1.34 per cycle and CU at
 * configs[3]'s size, a quarter of them per-piece bookkeeping; profiles/r06_k5_counters.txt). */
struct __attribute__((aligned(16))) wga_cov_desc {
  u32 g;  /* tile */
  u32 ab; /* first op | end op << 12 | (where the loads end - end op) << 24 | WGA_COV_WIDE */
  u32 rb; /* 4 x (coverage index in front of op `first` - window start), two's complement; WIDE: the piece's index, low half */
  u32 lc; /* 4 x min(limit - window start, WGA_COV_WIN); WIDE: the index's high half | bit 31: in the list regions, not the tile slots */
};
#define WGA_COV_WIDE (1u << 31)
/* The list pass writes a tile's first WGA_COV_TILE_CAP pieces into the tile's own slots — no atomic with an answer to wait for —
 * and further ones where WGA_COV_LISTS counters hand out places (tile g uses counter g mod WGA_COV_LISTS: one counter for all
 * tiles would take every such segment of the batch through one address), each over a region of `rcap` pieces. */
#ifndef WGA_COV_TILE_CAP
#define WGA_COV_TILE_CAP (WGA_COV_LO / 2u) /* 8 slots per 1 024 ops */
#endif
#define WGA_COV_LISTS 4096u
#define WGA_COV_READY (1ull << 63)
/* inclusive scan over the lanes of a value below 2^40 (a lane's 16 ops advance less than 16 x 2^28), and the wave's total: two
 * 32-bit DPP scans, on the low 24 bits and on the rest */
__device__ __forceinline__ u64 cov_incl_scan_u64(u64 v, u64& total) {
  const u32 lo = wave_incl_scan_u32((u32)v & 0xFFFFFFu), hi = wave_incl_scan_u32((u32)(v >> 24));
  total = (u64)wave_last_u32(lo) + ((u64)wave_last_u32(hi) << 24);
  return (u64)lo + ((u64)hi << 24);
}

/* WGA_COV_LO consecutive ops per lane of tile g, zeros from op nt on: every 16-byte group that starts in front of nt is loaded (the group
 * that holds the stream's last op may reach up to 12 bytes beyond it, inside the same aligned 16 bytes; the callers give what
 * it brings from there no weight — the list pass only looks at ops inside record segments) */
__device__ __forceinline__ void cov_load_ops(const u32* __restrict__ ops, u64 tile_start, u32 nt, u32 lane,
                                             u32 w[WGA_COV_LO]) {
#pragma unroll
  for (u32 j = 0; j < WGA_COV_LO / 4u; j++) {
    const u32 base = lane * WGA_COV_LO + j * 4u;
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
/* bit c set: an op of code c moves on the target / is not counted (all but M and =).  The 16 bits stand twice in the word: v_bfe_i32
 * takes its bit index from the low FIVE bits of the operand, so bit_mask(BITS, op) works on the packed op itself — bit 4 is the
 * length's lowest bit and picks one copy or the other — without an instruction to cut the code out first */
#define WGA_COV_MOVES_BITS 0xFDEDFDEDu
#define WGA_COV_NOTCNT_BITS 0xFF7EFF7Eu
/* target advance of this lane's ops inside [a, b); mvl[e] = the advance of op e */
__device__ __forceinline__ u64 cov_lane_moves(const u32 mvl[WGA_COV_LO], u32 lane, u32 a, u32 b) {
  u64 mv = 0; /* 32 lengths below 2^28 do not fit 32 bits */
  const u32 t = lane * WGA_COV_LO - a, n = b - a;
#pragma unroll
  for (u32 e = 0; e < WGA_COV_LO; e++) mv += (t + e < n) ? mvl[e] : 0u;
  return mv;
}

/* Target advance of record r's ops in front of tile g (the record starts at op rs, in tile g0 = rs / WGA_COV_TILE): the tail sums the
 * tiles g0 .. g-1 published (every one of them ends inside the record, so its last segment is the record's part of it).  Lane L
 * takes tile g-1-L (and 64 further back per round); `early` is what a first poll of round 0 — sent before the tile's other
 * segments were worked on — brought back (cov_poll_early).  Those tiles belong to blocks of the same launch with lower
 * indices, which were dispatched before this one; should one of them not have published after `spin_limit` polls (or with a
 * limit of 0: at once), the ops themselves are added up — the pass ends whatever the dispatch order is. */
__device__ __forceinline__ u64 cov_poll_early(u64* tile_tail, u64 rs, u64 g, u32 lane) {
  const u64 g0 = rs >> WGA_COV_TILE_SHIFT;
  return (g - g0 > (u64)lane) ? (u64)atomicAdd((unsigned long long*)&tile_tail[g - 1 - lane], 0ull) : 0ull;
}
__device__ __forceinline__ u64 cov_look_back(u64* tile_tail, const u32* __restrict__ ops, u64 rs, u64 g, u32 lane,
                                             u32 spin_limit, u64 early) {
  const u64 g0 = rs >> WGA_COV_TILE_SHIFT;
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
  for (u64 i = rs + lane; i < (g << WGA_COV_TILE_SHIFT); i += 64) {
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
/* One thread per record: the tiles whose first op the record holds get the record's data (a record of 1 300 ops holds 1.3 tiles'
 * first ops; one without ops none).  A record that holds more than 16 has the whole wave write them.  (Until round 6 a thread per
 * tile looked its record up by bisection: 25 dependent loads each, 1.04 ms at configs[3]'s size.) */
__global__ __launch_bounds__(256) void k_cov_tile_info(const u64* __restrict__ op_off, u32 n, u64 n_ops,
                                                       const wga_cov_rec* __restrict__ rec_pos, wga_cov_tile* __restrict__ info) {
  const u32 r = blockIdx.x * 256u + threadIdx.x;
  const u32 lane = threadIdx.x & 63u;
  wga_cov_tile t;
  t.rec = r, t.pad = 0, t.rs = 0, t.re = 0, t.re1 = 0;
  t.rp0.pos0 = t.rp0.limit = t.rp1.pos0 = t.rp1.limit = 0;
  u64 g0 = 1, g1 = 0;
  if (r < n) {
    t.rs = op_off[r];
    t.re = op_off[r + 1];
    if (t.re > t.rs) { /* the tiles g with rs <= g * WGA_COV_TILE < re */
      g0 = (t.rs + (u64)(WGA_COV_TILE - 1u)) >> WGA_COV_TILE_SHIFT;
      g1 = (t.re - 1ull) >> WGA_COV_TILE_SHIFT;
    }
    if (g0 <= g1) {
      const bool more = r + 1u < n;
      t.rp0 = rec_pos[r];
      t.re1 = more ? op_off[r + 2] : 0ull;
      t.rp1 = rec_pos[more ? r + 1u : r];
    }
  }
  const u64 cnt = g0 <= g1 ? g1 - g0 + 1ull : 0ull;
  if (cnt <= 16ull)
    for (u64 g = g0; g <= g1; g++) info[g] = t;
  for (u64 big = __ballot(cnt > 16ull); big; big &= big - 1ull) { /* wave-uniform */
    const int src = __ffsll((unsigned long long)big) - 1;
    wga_cov_tile b;
    b.rs = __shfl(t.rs, src), b.re = __shfl(t.re, src), b.re1 = __shfl(t.re1, src);
    b.rp0.pos0 = __shfl(t.rp0.pos0, src), b.rp0.limit = __shfl(t.rp0.limit, src);
    b.rp1.pos0 = __shfl(t.rp1.pos0, src), b.rp1.limit = __shfl(t.rp1.limit, src);
    b.rec = __shfl(t.rec, src), b.pad = 0;
    const u64 f = __shfl(g0, src), l = __shfl(g1, src);
    for (u64 g = f + lane; g <= l; g += 64) info[g] = b;
  }
}

/* The list pass: one wave per tile of 1024 ops.  Every record segment of the tile is measured (target advance per lane, scanned),
 * the tile's last segment is published for the tiles behind it, the first segment looks back for where its record stands, and
 * every (segment, window) piece is counted under its window (win_cnt) and written to the tile's slots (beyond WGA_COV_TILE_CAP
 * pieces: to the tile's list region) — k_cov_place_* take them to their windows.  Segments other
 * than the first start with their record, so only the first one waits.  The order is: publish, look back, write.  What the first two
 * segments need of their records comes with the tile's ops in one load (k_cov_tile_info), a further segment's record data
 * (op_off, k_cov_rec_pos's pair) is fetched a segment ahead.  With `rcap` = 0 the pieces beyond the slots are only counted. */
#ifndef WGA_K5_LIST_WAVES
#define WGA_K5_LIST_WAVES 1 /* waves per SIMD the register allocation aims at (1: as many registers as it likes — 72, seven waves; with 32 ops per lane 4: 128 VGPRs) */
#endif
/* one tile of the list pass; w = the tile's packed ops, WGA_COV_LO consecutive ones per lane */
__device__ __forceinline__ void cov_list_tile(
    const u64 g, u32 (&w)[WGA_COV_LO], u64* const s_tail, const u32 lane, const u32* __restrict__ ops,
    const u64* __restrict__ op_off, u64 n_ops,
    const wga_cov_tile* __restrict__ tile_info, const wga_cov_rec* __restrict__ rec_pos, u64* tile_tail, u32* win_cnt,
    wga_cov_piece* tile_list, u32* tile_cnt, u64* list_cnt, wga_cov_piece* list, u64 rcap, u32 spin_limit) {
  const u64 tile_start = g << WGA_COV_TILE_SHIFT;
  const u64 tile_end = tile_start + WGA_COV_TILE < n_ops ? tile_start + WGA_COV_TILE : n_ops;
  const u32 nt = (u32)(tile_end - tile_start);
  const wga_cov_tile tr = tile_info[g];
  const u64 rs_next = tile_end < n_ops ? tile_info[g + 1].rs : ~0ull; /* where the record of the next tile's first op starts */
#pragma unroll
  for (u32 e = 0; e < WGA_COV_LO; e++) w[e] = (w[e] >> 4) & bit_mask(WGA_COV_MOVES_BITS, w[e]); /* the pass needs the ops' advance only */
  if (nt & 3u) { /* wave-uniform, the stream's last tile: what the last 16-byte group brought from behind the stream */
#pragma unroll
    for (u32 e = 0; e < WGA_COV_LO; e++) w[e] = (int)e < (int)(nt - lane * WGA_COV_LO) ? w[e] : 0u;
  }
  u32 any = 0; /* a lane's running sums stay below 2^31 when none of its ops advances 2^26 bases or more */
#pragma unroll
  for (u32 e = 0; e < WGA_COV_LO; e++) any |= w[e];
  const bool small_ops = __ballot(any >= (1u << 26)) == 0ull; /* wave-uniform */
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
      const u32 a2 = a > WGA_COV_LO * l1 ? a : WGA_COV_LO * l1, b2 = b < WGA_COV_LO * (l2 + 1u) ? b : WGA_COV_LO * (l2 + 1u);
      const u64 pos_a2 = __shfl(l_start, (int)l1);
      if (on) {
        atomicAdd(&win_cnt[wi], 1u);
        wga_cov_piece pc;
        pc.g = (u32)g;
        pc.ab = a2 | (b2 << 16);
        pc.pos0 = pos_a2;
        pc.limit = rp.limit;
        pc.wi = (u32)wi;
        pc.pad = span < (1ull << WGA_COV_NARROW_BITS) ? WGA_COV_NARROW : 0u; /* the replay may walk it in 32-bit window positions */
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
  /* The first segment's record began in a tile in front.  NEAR — in a tile of THIS block (the tile right in front, four tiles of
   * five on configs[3]'s records; since round 6 also two or three tiles in front, a quarter of its records: the tiles between
   * belong to the record from end to end, so their published sums are whole-tile sums) — its sum comes through LDS behind a block barrier
   * (hundreds of cycles) instead of through memory (a poll is a round trip or two of microseconds: the look-back was 7.4 of the
   * pass's 28 ms, profiles/r05_k5_list_pass_ablations.txt).  The block's first wave, and records that began further back, go
   * through the published sums of the tiles in front as before. */
  const u32 wave = WGA_WAVE_ID(threadIdx.x);
  const bool waits = rs0 < tile_start;
  const u64 back = g - (rs0 >> WGA_COV_TILE_SHIFT); /* the record began that many tiles in front (waits: >= 1) */
  const bool near_lds = waits && back <= (u64)wave; /* ... in a tile of this block: every tile between is the record's from end to end */
  const u64 early = waits && !near_lds ? cov_poll_early(tile_tail, rs0, g, lane) : 0ull;
  /* ONE scan serves every segment of the tile: the lanes' advance summed over all 1 024 ops, whatever records they belong to.
   * The advance in front of op i (wave-uniform i) is the sum of the lanes in front of i's lane plus that lane's ops in front of
   * i: a handful of adds in every lane and two v_readlane.  A tile that advances less than 2^31 bases (NARROW: every tile of a
   * real alignment) is then handled in 32-bit positions relative to each segment's start; anything else, a segment across more
   * than the slots' windows or a tile of more than WGA_COV_TILE_CAP pieces takes the general walk below, which measures every
   * segment on its own (64-bit scans, a lane-per-window search). */
  /* w becomes the running sum inside the lane (w[e] = advance of the lane's ops 0 .. e): the advance in front of any op is then
   * one register of one lane, read with a wave-uniform register index */
#pragma unroll
  for (u32 e = 1; e < WGA_COV_LO; e++) w[e] += w[e - 1]; /* modulo 2^32: exact when small_ops, and exactly undone below otherwise */
  const u32 lt = w[WGA_COV_LO - 1u];
  auto per_op_again = [&]() { /* the general walk measures ops one by one */
#pragma unroll
    for (u32 e = WGA_COV_LO - 1u; e > 0u; e--) w[e] -= w[e - 1];
  };
  /* one 32-bit scan when no lane's ops advance 2^25 bases (64 lanes stay below 2^31: every tile of a real alignment) */
  u64 tile_total;
  u32 Pin;
  if (small_ops && __ballot(lt >= (1u << 25)) == 0ull) { /* wave-uniform */
    Pin = wave_incl_scan_u32(lt);
    tile_total = (u64)wave_last_u32(Pin);
  } else {
    Pin = (u32)cov_incl_scan_u64((u64)lt, tile_total);
  }
  const bool narrow = small_ops && tile_total < (1ull << 31); /* wave-uniform */
  const u32 Pex = Pin - lt;
  auto prefix_at = [&](u32 i) -> u32 { /* NARROW only; i <= nt, wave-uniform */
    if (i >= WGA_COV_TILE) return (u32)tile_total;
    const u32 li = i >> WGA_COV_LO_SHIFT, e = WGA_UNI32(i & (WGA_COV_LO - 1u));
    const u32 part = e ? w[(e - 1u) & (WGA_COV_LO - 1u)] : 0u;
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
    if (lane == 0) s_tail[wave] = span_last;
  }
  __syncthreads(); /* every wave of the block comes here (k_cov_list_pieces): the waves' sums are in LDS */
  u64 base0 = 0;
  if (near_lds) { /* wave-uniform: the sums of the block's tiles from the record's first one on */
    for (u32 k = 1; k <= (u32)back; k++) base0 += WGA_UNI64(s_tail[wave - k]);
  } else if (waits) {
    base0 = cov_look_back(tile_tail, ops, rs0, g, lane, spin_limit, early);
  }
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
      const u32 pad = span < (1u << WGA_COV_NARROW_BITS) ? WGA_COV_NARROW : 0u;
      for (u32 j = 0; j < np; j++) { /* a window replays only the lanes that can mark inside it */
        u32 l1 = (u32)__popcll(__ballot(wl < j)); /* lanes that end in front of the window */
        l1 = l1 < 63u ? l1 : 63u;
        const u32 c2 = (u32)__popcll(__ballot(wf <= j)); /* lanes that start inside or in front of it */
        const u32 l2 = c2 ? c2 - 1u : 0u;
        const u32 a2 = a > WGA_COV_LO * l1 ? a : WGA_COV_LO * l1, b2 = b < WGA_COV_LO * (l2 + 1u) ? b : WGA_COV_LO * (l2 + 1u);
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

/* One wave per tile; a block's WGA_K5_LIST_BW waves take consecutive tiles, so that a wave finds the sum of the tile in front
 * of its own in LDS (cov_list_tile): 27.95 -> 23.3 ms at configs[3]'s size with four waves per block (eight: 28.9 ms, sixteen:
 * 39.0 — the barrier holds more waves; four with the blocks of one XCD on one contiguous eighth of the tiles: 33.0;
 * profiles/r05_k5_stated_run8_lds_handover_variants.txt).
 * Measured against one wave per tile with every look-back through memory, all within 1 ms of its 28 ms or slower: a grid of
 * resident waves that take every W-th tile, with the next tile's ops requested early (29.4 ms at five waves per SIMD) and
 * without (28.0 at seven, 28.9 at five); tiles of 2 048 ops (31.8); a third fewer vector instructions per tile (646 -> 403,
 * same time); the wave reading the tile in front again to add up a near record's ops itself (38.4 ms: the second read of a
 * line another wave of the same CU has just asked for goes to HBM again).  The ablations of round 5
 * (profiles/r05_k5_list_pass_ablations.txt: 27.95 ms as it was; without the window counts 27.2, without the piece store 26.3,
 * WITHOUT THE LOOK-BACK 20.5) said where the time went.
 * Round 6, on the tree with 16 K windows (22.6 ms as it stands, gpurun_out/r06lds): the barrier replaced by a word per wave in LDS
 * that the next wave polls (cleared behind a barrier at the block's start) 24.0 ms, with eight waves per block 30.3, sixteen 38.7
 * — larger blocks lose to the CU's 28 wave slots (one block of sixteen fits), not to the barrier; eight waves per SIMD (63
 * registers, 14 scalar spills) 24.7; tiles of 2 048 ops 28.7-31.5; a fifth fewer vector instructions (v_bfe_i32 on the packed
 * op, one scan instead of two) 0.4 ms.  The pass moves 116 GB at 5.0 TB/s.
 * Timing-only builds with parts left out (gpurun_out/k5abl, k5abl2; 22.8 ms as it stood): without the look-back through memory
 * 20.2; nothing but loads, scan and barrier 19.8-20.3; the same with every load one contiguous kilobyte instead of a lane's own
 * sixteen ops 16.4 — but the real pass with contiguous loads turned into the lane layout through LDS (swizzled, conflict-free)
 * ran 24.4 against 22.1: the round trip through LDS stands in every wave's critical path.  Kept from that series: a record
 * that began two or three tiles in front INSIDE the block takes its position from LDS as well (22.8 -> 22.3). */
#ifndef WGA_K5_LIST_BW
#define WGA_K5_LIST_BW 4u /* waves per block */
#endif
__global__ __launch_bounds__(64 * WGA_K5_LIST_BW, WGA_K5_LIST_WAVES) void k_cov_list_pieces(
    const u32* __restrict__ ops, const u64* __restrict__ op_off, u64 n_ops, const wga_cov_tile* __restrict__ tile_info,
    const wga_cov_rec* __restrict__ rec_pos, u64* tile_tail, u32* win_cnt, wga_cov_piece* tile_list, u32* tile_cnt, u64* list_cnt,
    wga_cov_piece* list, u64 rcap, u32 spin_limit) {
  __shared__ u64 s_tail[WGA_K5_LIST_BW];
  const u32 lane = threadIdx.x & 63u;
  const u64 g = (u64)blockIdx.x * WGA_K5_LIST_BW + WGA_WAVE_ID(threadIdx.x);
  const u64 tile_start = g << WGA_COV_TILE_SHIFT;
  if (tile_start >= n_ops) { /* the waves behind the stream's last tile only keep the block's barrier company */
    __syncthreads();
    return;
  }
  u32 w[WGA_COV_LO];
  cov_load_ops(ops, tile_start, tile_start + WGA_COV_TILE < n_ops ? WGA_COV_TILE : (u32)(n_ops - tile_start), lane, w);
  cov_list_tile(g, w, s_tail, lane, ops, op_off, n_ops, tile_info, rec_pos, tile_tail, win_cnt, tile_list, tile_cnt, list_cnt, list, rcap,
                spin_limit);
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
#ifndef WGA_COV_WAVES
#define WGA_COV_WAVES 16u
#endif
#define WGA_COV_BLOCK (64u * WGA_COV_WAVES)
#ifndef WGA_COV_AHEAD
#define WGA_COV_AHEAD 1
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
  const u64 tile_start = (u64)pc.g << WGA_COV_TILE_SHIFT;
  const u32 nt = tile_start + WGA_COV_TILE < n_ops ? WGA_COV_TILE : (u32)(n_ops - tile_start);
  const u32 b4 = ((pc.ab >> 16) + 3u) & ~3u;
  return b4 < nt ? b4 : nt;
}
/* the replay's descriptor of a listed piece; `idx` / `in_list` say where the piece itself stands (a WIDE one is read from there) */
__device__ __forceinline__ wga_cov_desc cov_make_desc(const wga_cov_piece& pc, u64 n_ops, u64 idx, bool in_list) {
  static_assert(WGA_COV_TILE < 4096u, "12-bit op indices");
  const u32 a = pc.ab & 0xFFFFu, b = pc.ab >> 16;
  wga_cov_desc d;
  d.g = pc.g;
  d.ab = a | (b << 12) | ((cov_piece_lim(n_ops, pc) - b) << 24);
  if (pc.pad & WGA_COV_NARROW) {
    const u64 w0 = (u64)pc.wi << WGA_COV_WIN_SHIFT;
    const u64 room = pc.limit - w0; /* the piece is listed under this window: limit > w0 */
    d.rb = (u32)(pc.pos0 - w0) << 2;
    d.lc = (room < (u64)WGA_COV_WIN ? (u32)room : WGA_COV_WIN) << 2;
  } else {
    d.ab |= WGA_COV_WIDE;
    d.rb = (u32)idx;
    d.lc = (u32)(idx >> 32) | (in_list ? 1u << 31 : 0u);
  }
  return d;
}
/* the listed pieces go to their windows: a piece takes the next place of its window (win_fill, zero before) — an atomic with an
 * answer per piece, but of threads that have nothing else to wait for — and leaves its descriptor there */
__global__ __launch_bounds__(256) void k_cov_place_tiles(u64 n_tiles, u64 n_ops, const u32* __restrict__ tile_cnt,
                                                         const wga_cov_piece* __restrict__ tile_list, u32* win_fill,
                                                         const u64* __restrict__ win_off, wga_cov_desc* descs) {
  const u64 t = (u64)blockIdx.x * WGA_BLOCK + threadIdx.x;
  const u64 g = t / WGA_COV_TILE_CAP;
  if (g >= n_tiles || (u32)(t % WGA_COV_TILE_CAP) >= tile_cnt[g]) return;
  const wga_cov_piece pc = tile_list[t];
  descs[win_off[pc.wi] + atomicAdd(&win_fill[pc.wi], 1u)] = cov_make_desc(pc, n_ops, t, false);
}
__global__ __launch_bounds__(256) void k_cov_place_pieces(u64 n_ops, const u64* __restrict__ list_cnt,
                                                          const wga_cov_piece* __restrict__ list, u64 rcap, u32* win_fill,
                                                          const u64* __restrict__ win_off, wga_cov_desc* descs) {
  const u32 region = blockIdx.y;
  const u64 i = (u64)blockIdx.x * WGA_BLOCK + threadIdx.x;
  if (i >= list_cnt[region]) return;
  const wga_cov_piece pc = list[(u64)region * rcap + i];
  descs[win_off[pc.wi] + atomicAdd(&win_fill[pc.wi], 1u)] = cov_make_desc(pc, n_ops, (u64)region * rcap + i, true);
}
/* the four ops of this lane in the first 256-op step of a piece */
__device__ __forceinline__ void cov_step_ops(const u32* __restrict__ ops, const wga_cov_desc& d, u32 lane, u32 (&w)[4]) {
  const u32 b = (d.ab >> 12) & 0xFFFu;
  cov_load4(ops, (u64)d.g << WGA_COV_TILE_SHIFT, b + ((d.ab >> 24) & 3u), ((d.ab & 0xFFFu) & ~3u) + lane * 4u, w);
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
      /* Forward progress rests on the launch's order: the window a block waits for has a lower rank in `order` (or a lower
       * index without one) and blocks are dispatched in blockIdx order, so it is running or done.  Should that ever not hold (a
       * partitioned or preempted device, a launch order that changes), the wait ends after ~4 s of polls in a trap — the call
       * then FAILS at the stream's next synchronisation instead of hanging the device. */
      if (polls > (1u << 24)) __builtin_trap();
    }
    const u32 first = pref ? (u32)__ffsll((unsigned long long)pref) - 1u : 64u;
    acc += wave_sum_u32((mine && lane <= first) ? (u32)v : 0u);
    if (pref) return acc;
  }
}

template <bool FINAL>
__global__ __launch_bounds__(WGA_COV_BLOCK, 8) void k_cov_windows(const u32* __restrict__ ops, u64 n_ops,
                                                               const wga_cov_desc* __restrict__ descs,
                                                               const wga_cov_piece* __restrict__ tile_list,
                                                               const wga_cov_piece* __restrict__ list,
                                                               const u64* __restrict__ win_off, int* cov, u64 n_cov,
                                                               const u64* __restrict__ rng_lo, const u64* __restrict__ rng_hi,
                                                               u32 n_rng, u64* win_state, const u32* __restrict__ order) {
  /* (round 6: one pad word per 32 counters against the count pass's bank conflicts — 43 % of its LDS cycles,
   * profiles/r05_k1_k5_counters.txt — measured at configs[3]'s stated size: 70.7 ms against 69.3-69.6 without.  LDS is not what the
   * pass waits for; the padding was taken out again.)
   * Behind the window's counters stands one word per lane: a mark that does not count (an op that is not M / =, a position
   * outside the window or beyond the target) is added THERE — every lane its own word, no conflict — instead of being
   * branched around: a compare, a select and the LDS add per mark where the exec-masked form spent seven instructions. */
  __shared__ int s_win[WGA_COV_WIN + 64u];
  __shared__ u32 s_ws[WGA_COV_WAVES + 1];
  __shared__ u32 s_wf[WGA_COV_WAVES];
  constexpr int D = WGA_COV_AHEAD;
  constexpr u32 PER = WGA_COV_WIN / WGA_COV_BLOCK;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  /* FINAL: the blocks take the windows in `order` — the windows that start a range (or lie outside every range) first, then every
   * range's second window, third ... — so that the windows in flight at one time are a few consecutive ones of MANY
   * ranges instead of hundreds of consecutive ones of one: a window only waits for the windows of its own range in front of it,
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
    wga_cov_desc dq[2 * D + 1];
    u32 wq[D + 1][4];
#pragma unroll
    for (int k = 0; k <= 2 * D; k++) {
      const u64 q = p0 + (u64)k * WGA_COV_WAVES;
      dq[k] = descs[q < p_hi ? q : p_lo];
    }
#pragma unroll
    for (int k = 0; k <= D; k++) {
      wq[k][0] = wq[k][1] = wq[k][2] = wq[k][3] = 0u;
      if (p0 + (u64)k * WGA_COV_WAVES < p_hi) cov_step_ops(ops, dq[k], lane, wq[k]);
    }
    const u32 dummy = (WGA_COV_WIN + lane) << 2;
    for (u64 p = p0; p < p_hi; p += WGA_COV_WAVES) {
      const wga_cov_desc pd = dq[0];
      u32 w[4] = {wq[0][0], wq[0][1], wq[0][2], wq[0][3]};
#pragma unroll
      for (int k = 0; k < 2 * D; k++) dq[k] = dq[k + 1];
#pragma unroll
      for (int k = 0; k < D; k++) {
#pragma unroll
        for (int e = 0; e < 4; e++) wq[k][e] = wq[k + 1][e];
      }
      if (p + (u64)(2 * D + 1) * WGA_COV_WAVES < p_hi) dq[2 * D] = descs[p + (u64)(2 * D + 1) * WGA_COV_WAVES];
      if (p + (u64)(D + 1) * WGA_COV_WAVES < p_hi) cov_step_ops(ops, dq[D], lane, wq[D]);
      const u64 tile_start = (u64)pd.g << WGA_COV_TILE_SHIFT;
      const u32 a = pd.ab & 0xFFFu, b = (pd.ab >> 12) & 0xFFFu;
      const u32 lim = b + ((pd.ab >> 24) & 3u);
      if (!(pd.ab & WGA_COV_WIDE)) { /* wave-uniform */
        const u32 Lc = pd.lc; /* byte offsets: a position in front of the window compares above Lc (two's complement) */
        /* FULLW: the target goes on behind the window (nearly every piece) — whatever lies at or beyond the window's end goes to
         * the dummy words with ONE v_min (it may land in another lane's word: they hold nothing) */
        auto walk = [&](auto fullw) {
          constexpr bool FULLW = decltype(fullw)::value;
          u32 rb = pd.rb;
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
            for (int e = 0; e < 4; e++) { /* 4 x the op's advance: the length stands at bit 4 */
              lm[e] = (w[e] >> 2) & (bit_mask(WGA_COV_MOVES_BITS, w[e]) << 2);
              mv += lm[e];
            }
            const u32 inc = wave_incl_scan_u32(mv);
            u32 r = rb + (inc - mv);
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const u32 nc = bit_mask(WGA_COV_NOTCNT_BITS, w[e]);
              const u32 ku = r | nc;
              r += lm[e]; /* an op that counts moves */
              const u32 kd = r | nc;
              const u32 iu = FULLW ? (ku < dummy ? ku : dummy) : (ku < Lc ? ku : dummy);
              const u32 id = FULLW ? (kd < dummy ? kd : dummy) : (kd < Lc ? kd : dummy);
              atomicAdd((int*)((char*)s_win + iu), 1);
              atomicAdd((int*)((char*)s_win + id), -1);
            }
            rb += wave_last_u32(inc);
#pragma unroll
            for (int e = 0; e < 4; e++) w[e] = wn[e];
          }
        };
        if (Lc == (WGA_COV_WIN << 2))
          walk(std::true_type());
        else
          walk(std::false_type());
      } else {
        const u64 src = (u64)pd.rb | ((u64)(pd.lc & 0x7FFFFFFFu) << 32);
        const wga_cov_piece pc = (pd.lc >> 31) ? list[src] : tile_list[src];
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

#endif /* WGA_K5_PAFCOV_H */
