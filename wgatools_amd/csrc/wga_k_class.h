/*
 * wga_k_class.h — class sums of the op stream per tile and per record (shared first pass of pafcov and pafpseudo), the stat totals.
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K_CLASS_H
#define WGA_K_CLASS_H

#include "wga_kernels.h"

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
  /* The sums as K1 makes them since round 6 (wga_kernels.h): a class sum is `len & mask`, the mask one v_bfe_i32 of a class
   * constant (its 16 bits standing twice) by the packed op itself; the fifth class is what the four others leave of the sum of
   * all lengths; the whole tile is summed first, without a range test, and a tile's last segment is what the segments in front
   * leave of it; one 32-bit reduction per sum when no lane's lengths reach 2^26.  14 vector instructions per op (17 with the
   * range test) where the class number, five compares and five selects took 45 (0.67 -> 0.3x ms on configs[1]'s batch). */
  constexpr u32 MX_BITS = 0x01810181u, I_BITS = 0x02020202u, D_BITS = 0x04040404u, S_BITS = 0x00100010u;
  auto wave_sums = [&](const u32 (&p)[5], u64 (&S)[5]) { /* p[4]: every op's length */
    u64 T;
    if (__ballot(p[4] >= (1u << 26)) == 0ull) { /* wave-uniform */
#pragma unroll
      for (int c = 0; c < 4; c++) S[c] = wave_sum_u32(p[c]);
      T = wave_sum_u32(p[4]);
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) S[c] = wave_sum_u32_wide(p[c]);
      T = wave_sum_u32_wide(p[4]);
    }
    S[4] = T - S[0] - S[1] - S[2] - S[3]; /* every op is M-like, I, D, S or other */
  };
  u64 W[5], A[5] = {0, 0, 0, 0, 0}; /* the whole tile; its segments so far */
  {
    u32 p[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const u32 op = w[k], len = op >> 4;
      p[0] += len & bit_mask(MX_BITS, op);
      p[1] += len & bit_mask(I_BITS, op);
      p[2] += len & bit_mask(D_BITS, op);
      p[3] += len & bit_mask(S_BITS, op);
      p[4] += len;
    }
    wave_sums(p, W);
  }
  while (cur < tile_end) {
    u64 re = op_off[r + 1];
    while (re <= cur) {
      r++;
      re = op_off[r + 1];
    }
    const u64 rs = op_off[r];
    const u64 seg_end = re < tile_end ? re : tile_end;
    const u32 a = (u32)(cur - tile_start), b = (u32)(seg_end - tile_start);
    u64 S[5];
    if (seg_end == tile_end) { /* wave-uniform */
#pragma unroll
      for (int c = 0; c < 5; c++) S[c] = W[c] - A[c];
    } else {
      const u32 span = b - a;
      u32 p[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const u32 idx = ((u32)(k >> 2) * 64u + lane) * 4u + (u32)(k & 3);
        const u32 op = (idx - a < span) ? w[k] : 0u;
        const u32 len = op >> 4;
        p[0] += len & bit_mask(MX_BITS, op);
        p[1] += len & bit_mask(I_BITS, op);
        p[2] += len & bit_mask(D_BITS, op);
        p[3] += len & bit_mask(S_BITS, op);
        p[4] += len;
      }
      wave_sums(p, S);
#pragma unroll
      for (int c = 0; c < 5; c++) A[c] += S[c];
    }
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
/* stat totals: the sum of all records' counters (what `stat` aggregates per pair, stat.rs:181-223, */
/* for one pair; the 88 bytes a multi-GPU run all-reduces)                                         */
/* ============================================================================================ */
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

#endif /* WGA_K_CLASS_H */
