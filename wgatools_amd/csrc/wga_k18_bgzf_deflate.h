/*
 * wga_k18_bgzf_deflate.h — K18: output bytes in HBM -> BGZF (blocked gzip) members, compressed on the device.
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K18_BGZF_DEFLATE_H
#define WGA_K18_BGZF_DEFLATE_H

#include "wga_kernels.h"

typedef int i32;
typedef unsigned short u16;

/* ============================================================================================ */
/* K18: `-o out.maf.gz` without 15 GB over PCIe and a host deflate behind it                    */
/* ============================================================================================ */
/* The reference writes `.gz` outputs through flate2's GzEncoder at level 6 (utils.rs:181-228): what a reader of the file
 * is promised is a gzip stream that inflates to the bytes the plain output would have held — not particular compressed
 * bytes.  This part keeps that promise with the text still in HBM: the byte stream is cut into members of 32 768 input
 * bytes, each a complete gzip member with the BGZF extra field (`BC`, total size - 1; SAM specification 4.1), so any gzip
 * reader takes the concatenation and htslib-style readers can seek in it.  A member's payload is ONE deflate block of
 * literals under a dynamic Huffman code built from the member's own byte histogram (no LZ77 matches: the match search is
 * the sequential part of deflate; alignment rows of four or five symbols come to 2.1-2.3 bits per base under their own
 * code, 3.5 x, which is what cuts the PCIe and page-cache bytes), or a stored block where that would not be smaller.
 *
 * Two launches of one workgroup (four waves) per member around one exclusive scan of the member sizes:
 *   k_bgzf_plan   stages the member's bytes in LDS while counting them (sixteen padded replicas of the histogram: the
 *                 four bases would otherwise meet in four LDS banks), CRC-32 of the bytes (each thread its 128 bytes by the
 *                 byte table, the 256 partial values folded pairwise by x^(8 len) mod P), code lengths by the two-queue
 *                 Huffman construction on the rank-sorted counts (counts halved and rebuilt while a length passes 15),
 *                 -> code lengths, CRC and member size.
 *   k_bgzf_emit   stages the bytes again, gives the lengths their canonical codes, packs header, literals and end-of-block
 *                 into an LDS image of the member that starts at the byte offset the member has inside its first aligned
 *                 output word, and writes that image with coalesced word stores (bytes at the two ragged ends).
 * Traffic: the input twice (the second time usually from the L2 / Infinity Cache for piece-sized calls), the output once,
 * 288 B of lengths per member.  The deflate header spells all 257 literal/length code lengths as 4-bit codes (a fixed
 * code-length code of sixteen 4-bit symbols): 139 bytes per member, 0.4 % of the input, for no run-length logic. */
#define WGA_BGZF_IN 32768u       /* input bytes per member */
#define WGA_BGZF_HDR 18u         /* gzip header with the BC extra field */
#define WGA_BGZF_TRAILER 8u      /* CRC-32, ISIZE */
#define WGA_BGZF_DYN_HDR_BITS 1106u /* 3 + 5 + 5 + 4 + 19 * 3 + 257 * 4 + 4 */
#define WGA_BGZF_OUT_WORDS 8216u /* 3 + 18 + (5 + 32768) + 8 bytes at most, in words, and room for a carry word */
#define WGA_BGZF_LENS 288u       /* bytes of code lengths kept per member between the launches (257 used) */
#define WGA_BGZF_REP 16u
#define WGA_BGZF_REP_STRIDE 129u /* words per histogram replica: two 16-bit counters per word, one word of padding */

struct wga_bgzf_member {
  u32 crc;
  u32 stored; /* 1: stored block */
};

/* CRC-32 (reflected 0xEDB88320) tables made at compile time: the byte table, and x^(2^k) mod P for the zero-feeding
 * products that fold partial CRCs (the same arithmetic as zlib's crc32_combine) */
struct wga_crc_tables {
  u32 byte[256];    /* the byte table */
  u32 by4[3][256];  /* its three companions for four bytes at a time (slicing-by-4: by4[k][i] = byte i followed by k + 1 zero bytes) */
  u32 x2n[32];
  static constexpr u32 mul(u32 a, u32 b) {
    u32 m = 1u << 31, p = 0;
    for (;;) {
      if (a & m) {
        p ^= b;
        if ((a & (m - 1u)) == 0u) break;
      }
      m >>= 1;
      b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
  }
  constexpr wga_crc_tables() : byte{}, by4{}, x2n{} {
    for (u32 i = 0; i < 256u; i++) {
      u32 c = i;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
      byte[i] = c;
    }
    for (u32 i = 0; i < 256u; i++) {
      u32 c = byte[i];
      for (int k = 0; k < 3; k++) by4[k][i] = c = byte[c & 0xFFu] ^ (c >> 8);
    }
    u32 p = 1u << 30; /* x^1 */
    x2n[0] = p;
    for (int k = 1; k < 32; k++) x2n[k] = p = mul(p, p);
  }
};
__device__ const wga_crc_tables k_crc_tables{};

__device__ __forceinline__ u32 crc_mul(u32 a, u32 b) {
  u32 p = 0;
  for (int k = 31; k >= 0; k--) { /* a's bit 31 is x^0 */
    p ^= (0u - ((a >> k) & 1u)) & b;
    b = (b >> 1) ^ ((0u - (b & 1u)) & 0xEDB88320u);
  }
  return p;
}
/* the CRC register after `n_bytes` zero bytes went through it (n_bytes < 2^29) */
__device__ __forceinline__ u32 crc_shift(u32 crc, u32 n_bytes) {
  u32 k = 3; /* x^(8 n) */
  while (n_bytes) {
    if (n_bytes & 1u) crc = crc_mul(k_crc_tables.x2n[k], crc);
    n_bytes >>= 1;
    k++;
  }
  return crc;
}

/* word `i` of a member's input (bytes 4 i .. 4 i + 3 behind `base`), read through aligned words whatever the alignment of
 * `base`; bytes at or behind `n` read as zero and no word without a byte of [base, base + n) is touched */
__device__ __forceinline__ u32 bgzf_in_word(const u8* base, u32 n, u32 i) {
  const u64 addr = (u64)base + 4ull * i;
  const u32 sh = (u32)(addr & 3ull) * 8u;
  const u32* a = (const u32*)(addr & ~3ull);
  const u64 end = (u64)base + n;
  u32 lo = 0, hi = 0;
  if ((u64)a < end && (u64)a + 4ull > (u64)base) lo = a[0];
  if (sh && (u64)(a + 1) < end) hi = a[1];
  u32 w = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
  const u32 at = 4u * i;
  if (at >= n) return 0u;
  if (n - at < 4u) w &= (1u << (8u * (n - at))) - 1u;
  return w;
}

/* OR `n` bits (n <= 32, value below 2^n) into the LDS bit stream at bit `pos` */
__device__ __forceinline__ void bgzf_put_bits(u32* s_out, u32 pos, u32 value, u32 n) {
  const u32 sh = pos & 31u, w = pos >> 5;
  atomicOr(&s_out[w], value << sh);
  if (sh + n > 32u) atomicOr(&s_out[w + 1u], value >> (32u - sh));
}

/* canonical codes of the lengths in s_len[0..256] -> s_code[s] = (bit-reversed code << 4) | length, by wave 0.
 * Symbol s of chunk k sits in lane s - 64 k; within a length the codes go up with the symbol (RFC 1951 3.2.2). */
__device__ __forceinline__ void bgzf_assign_codes(const u8* s_len, u32* s_code, u32 lane) {
  u32 len[5], rank[5];
  u32 count[16];
#pragma unroll
  for (int L = 0; L < 16; L++) count[L] = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const u32 s = (u32)k * 64u + lane;
    len[k] = s <= 256u ? (u32)s_len[s] : 0u;
    rank[k] = 0;
#pragma unroll
    for (int L = 1; L < 16; L++) {
      const u64 m = __ballot(len[k] == (u32)L);
      if (len[k] == (u32)L) rank[k] = count[L] + lane_rank(m, lane);
      count[L] += (u32)__popcll(m);
    }
  }
  u32 next[16];
  u32 code = 0;
  next[0] = 0;
#pragma unroll
  for (int L = 1; L < 16; L++) {
    code = (code + (L == 1 ? 0u : count[L - 1])) << 1;
    next[L] = code;
  }
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const u32 s = (u32)k * 64u + lane;
    if (s > 256u) continue;
    u32 e = 0;
    if (len[k]) {
      u32 first = 0;
#pragma unroll
      for (int L = 1; L < 16; L++) first = len[k] == (u32)L ? next[L] : first;
      e = ((__brev(first + rank[k]) >> (32u - len[k])) << 4) | len[k];
    }
    s_code[s] = e;
  }
}

__global__ __launch_bounds__(256) void k_bgzf_plan(const u8* __restrict__ in, u64 n_bytes, u64* __restrict__ member_bytes,
                                                   wga_bgzf_member* __restrict__ members, u8* __restrict__ lens) {
  __shared__ u32 s_in[WGA_BGZF_IN / 4u];
  __shared__ u32 s_rep[WGA_BGZF_REP * WGA_BGZF_REP_STRIDE];
  __shared__ u32 s_crc_t[4][256];
  __shared__ u32 s_cnt[260];   /* the member's histogram; [256] = the end-of-block symbol */
  __shared__ u32 s_wt[260];    /* the weights the code is built on (the counts, halved while a length passes 15) */
  __shared__ u32 s_leaf[260];  /* weights of the used symbols in rank order */
  __shared__ u32 s_node[260];  /* weights of the internal nodes in the order they are made */
  __shared__ u32 s_key[260];   /* (weight << 9) | symbol of the used symbols, in symbol order */
  __shared__ u16 s_par[520];   /* parent (index among the internal nodes) of leaf i, of internal node n + j */
  __shared__ u8 s_len[260];
  __shared__ u32 s_fold[4];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  const u64 b = blockIdx.x;
  const u8* base = in + b * (u64)WGA_BGZF_IN;
  const u64 left = n_bytes - b * (u64)WGA_BGZF_IN;
  const u32 n = left < (u64)WGA_BGZF_IN ? (u32)left : WGA_BGZF_IN;

  for (u32 i = tid; i < WGA_BGZF_REP * WGA_BGZF_REP_STRIDE; i += 256u) s_rep[i] = 0u;
  s_crc_t[0][tid] = k_crc_tables.byte[tid];
  for (int k = 0; k < 3; k++) s_crc_t[k + 1][tid] = k_crc_tables.by4[k][tid];
  __syncthreads();
  /* stage and count: word i of the member by thread i mod 256 */
  {
    u32* rep = s_rep + (lane & (WGA_BGZF_REP - 1u)) * WGA_BGZF_REP_STRIDE;
    for (u32 i = tid; i < WGA_BGZF_IN / 4u; i += 256u) {
      const u32 w = bgzf_in_word(base, n, i);
      s_in[i] = w;
      const u32 have = 4u * i < n ? (n - 4u * i < 4u ? n - 4u * i : 4u) : 0u;
      for (u32 j = 0; j < have; j++) {
        const u32 c = (w >> (8u * j)) & 0xFFu;
        atomicAdd(&rep[c >> 1], 1u << (16u * (c & 1u)));
      }
    }
  }
  __syncthreads();
  /* the histogram out of its replicas (a replica's counter holds at most 16 threads x 128 bytes) */
  for (u32 s = tid; s < 257u; s += 256u) {
    u32 c = 0;
    if (s < 256u)
      for (u32 r = 0; r < WGA_BGZF_REP; r++) c += (s_rep[r * WGA_BGZF_REP_STRIDE + (s >> 1)] >> (16u * (s & 1u))) & 0xFFFFu;
    else
      c = 1u;
    s_cnt[s] = c;
    s_wt[s] = c;
  }
  /* CRC-32: thread t takes the 128 bytes that END 128 (255 - t) bytes in front of the member's end (a short member has
   * its missing bytes in FRONT, where zero bytes leave a zero register alone), register 0 going in */
  u32 crc = 0;
  {
    const i32 start = (i32)n - 128 * (i32)(256u - tid);
    if ((n & 3u) == 0u) { /* four bytes a step (every member but a ragged last one) */
      for (i32 j = 0; j < 32; j++) {
        const i32 wi = (start >> 2) + j;
        if (wi < 0) continue;
        const u32 x = crc ^ s_in[wi];
        crc = s_crc_t[3][x & 0xFFu] ^ s_crc_t[2][(x >> 8) & 0xFFu] ^ s_crc_t[1][(x >> 16) & 0xFFu] ^ s_crc_t[0][x >> 24];
      }
    } else {
      const u8* bytes = (const u8*)s_in;
      for (i32 j = 0; j < 128; j++) {
        const i32 p = start + j;
        if (p >= 0) crc = s_crc_t[0][(crc ^ (u32)bytes[p]) & 0xFFu] ^ (crc >> 8);
      }
    }
  }
  /* fold: (a, b) -> a x^(8 |b|) + b, |b| = 128, 256, ... bytes */
  for (u32 lvl = 0; lvl < 6u; lvl++) {
    const u32 other = __shfl_up(crc, 1u << lvl);
    if ((lane & ((2u << lvl) - 1u)) == (2u << lvl) - 1u) crc = crc_mul(k_crc_tables.x2n[10u + lvl], other) ^ crc;
  }
  if (lane == 63u) s_fold[wave] = crc;
  __syncthreads();

  if (wave == 0u) {
    if (lane == 0u) {
      const u32 ab = crc_mul(k_crc_tables.x2n[16], s_fold[0]) ^ s_fold[1]; /* 64 threads x 128 bytes = 2^13 bytes = x^(2^16) */
      const u32 cd = crc_mul(k_crc_tables.x2n[16], s_fold[2]) ^ s_fold[3];
      const u32 all = crc_mul(k_crc_tables.x2n[17], ab) ^ cd;
      members[b].crc = all ^ crc_shift(0xFFFFFFFFu, n) ^ 0xFFFFFFFFu;
    }
    /* code lengths */
    for (;;) {
      /* rank of every used symbol by (weight, symbol) */
      u32 key[5], rank[5];
      u32 n_used = 0;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const u32 s = (u32)k * 64u + lane;
        const u32 wv = s <= 256u ? s_wt[s] : 0u;
        key[k] = wv ? (wv << 9) | s : 0u;
        rank[k] = 0;
        const u64 m = __ballot(key[k] != 0u);
        if (key[k]) s_key[n_used + lane_rank(m, lane)] = key[k];
        n_used += (u32)__popcll(m);
      }
      WGA_WAVE_SYNC();
      for (u32 o = 0; o < n_used; o++) {
        const u32 ko = s_key[o];
#pragma unroll
        for (int k = 0; k < 5; k++) rank[k] += ko < key[k] ? 1u : 0u;
      }
#pragma unroll
      for (int k = 0; k < 5; k++)
        if (key[k]) s_leaf[rank[k]] = key[k] >> 9;
      WGA_WAVE_SYNC();
      /* two queues: the leaves in rank order, the internal nodes in the order they were made (their weights never go
       * down); a leaf wins a tie, which keeps the tree as shallow as the weights allow */
      if (lane == 0u) {
        u32 a = 0, q = 0;
        u32 la = s_leaf[0], nq = 0xFFFFFFFFu; /* the two fronts; all ones: that queue has nothing to give right now */
        for (u32 j = 0; j + 1u < n_used; j++) {
          u32 wsum = 0;
          for (int pick = 0; pick < 2; pick++) {
            if (la <= nq) {
              wsum += la;
              s_par[a++] = (u16)j;
              la = a < n_used ? s_leaf[a] : 0xFFFFFFFFu;
            } else {
              wsum += nq;
              s_par[n_used + q++] = (u16)j;
              nq = q < j ? s_node[q] : 0xFFFFFFFFu;
            }
          }
          s_node[j] = wsum;
          if (q == j) nq = wsum; /* the node just made is the internal queue's front */
        }
      }
      WGA_WAVE_SYNC();
      /* a leaf's length = the nodes on its way up to the root (internal node n_used - 2) */
      u32 deepest = 0;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const u32 s = (u32)k * 64u + lane;
        if (s > 256u) continue;
        u32 l = 0;
        if (key[k]) {
          u32 p = s_par[rank[k]];
          l = 1;
          while (p != n_used - 2u) {
            p = s_par[n_used + p];
            l++;
          }
        }
        s_len[s] = (u8)(l > 255u ? 255u : l);
        deepest = l > deepest ? l : deepest;
      }
      const bool too_deep = __ballot(deepest > 15u) != 0ull;
      WGA_WAVE_SYNC();
      if (!too_deep) break;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const u32 s = (u32)k * 64u + lane;
        if (s <= 256u && s_wt[s]) s_wt[s] = (s_wt[s] + 1u) >> 1;
      }
      WGA_WAVE_SYNC();
    }
    /* the member's size under this code, and the stored block where that is not smaller */
    u32 bits = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const u32 s = (u32)k * 64u + lane;
      if (s <= 256u) bits += s_cnt[s] * (u32)s_len[s];
    }
    bits = wave_sum_u32(bits) + WGA_BGZF_DYN_HDR_BITS;
    const u32 dyn = (bits + 7u) >> 3, sto = 5u + n;
    const bool stored = dyn >= sto;
    if (lane == 0u) {
      members[b].stored = stored ? 1u : 0u;
      member_bytes[b] = (u64)(WGA_BGZF_HDR + (stored ? sto : dyn) + WGA_BGZF_TRAILER);
    }
    u8* out = lens + b * (u64)WGA_BGZF_LENS;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const u32 s = (u32)k * 64u + lane;
      if (s < WGA_BGZF_LENS) out[s] = s <= 256u ? s_len[s] : (u8)0;
    }
  }
}

__global__ __launch_bounds__(256) void k_bgzf_emit(const u8* __restrict__ in, u64 n_bytes, const u64* __restrict__ member_off,
                                                   const wga_bgzf_member* __restrict__ members, const u8* __restrict__ lens,
                                                   u8* __restrict__ out) {
  __shared__ u32 s_in[WGA_BGZF_IN / 4u];
  __shared__ u32 s_out[WGA_BGZF_OUT_WORDS];
  __shared__ u32 s_code[260];
  __shared__ u8 s_len[WGA_BGZF_LENS];
  __shared__ u64 s_w[5];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  const u64 b = blockIdx.x;
  const u8* base = in + b * (u64)WGA_BGZF_IN;
  const u64 left = n_bytes - b * (u64)WGA_BGZF_IN;
  const u32 n = left < (u64)WGA_BGZF_IN ? (u32)left : WGA_BGZF_IN;
  const u64 off = member_off[b];
  const u32 msize = (u32)(member_off[b + 1u] - off);
  const wga_bgzf_member mb = members[b];
  u8* dst = out + off;
  const u32 a = (u32)((u64)dst & 3ull); /* the member's first byte inside its first aligned output word */

  for (u32 i = tid; i < WGA_BGZF_IN / 4u; i += 256u) s_in[i] = bgzf_in_word(base, n, i);
  for (u32 i = tid; i < WGA_BGZF_OUT_WORDS; i += 256u) s_out[i] = 0u;
  for (u32 i = tid; i < WGA_BGZF_LENS; i += 256u) s_len[i] = lens[b * (u64)WGA_BGZF_LENS + i];
  __syncthreads();
  if (wave == 0u) bgzf_assign_codes(s_len, s_code, lane);
  __syncthreads();

  const u32 bit0 = 8u * a;                   /* the member's first bit in the image */
  const u32 pay0 = bit0 + 8u * WGA_BGZF_HDR; /* the deflate stream's first bit */
  const u8* bytes = (const u8*)s_in;
  /* gzip header: 1f 8b, deflate, FEXTRA, mtime 0, xfl 0, OS unknown, XLEN 6, 'B' 'C' 2 0, BSIZE */
  if (tid < 5u) {
    const u32 bsize = msize - 1u;
    const u32 h[5] = {0x04088B1Fu, 0x00000000u, 0x0006FF00u, 0x00024342u, bsize & 0xFFFFu};
    bgzf_put_bits(s_out, bit0 + 32u * tid, h[tid], tid < 4u ? 32u : 16u);
  }
  u32 pay_bytes;
  if (mb.stored) {
    pay_bytes = 5u + n;
    if (tid == 0u) {
      bgzf_put_bits(s_out, pay0, 1u, 8u); /* BFINAL = 1, BTYPE = 00, padding */
      bgzf_put_bits(s_out, pay0 + 8u, (n & 0xFFFFu) | ((~n & 0xFFFFu) << 16), 32u);
    }
    for (u32 i = tid; 4u * i < n; i += 256u) {
      const u32 have = n - 4u * i < 4u ? n - 4u * i : 4u;
      bgzf_put_bits(s_out, pay0 + 40u + 32u * i, s_in[i], 8u * have);
    }
  } else {
    /* header of the dynamic block */
    if (tid == 0u) {
      bgzf_put_bits(s_out, pay0, 1u | (2u << 1) | (0u << 3) | (0u << 8) | (15u << 13), 17u); /* BFINAL, BTYPE 10, HLIT 0, HDIST 0, HCLEN 15 */
      /* code-length code lengths in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15: 0 0 0 and sixteen 4s */
      for (u32 k = 3; k < 19u; k++) bgzf_put_bits(s_out, pay0 + 17u + 3u * k, 4u, 3u);
    }
    for (u32 s = tid; s <= 257u; s += 256u) {
      /* length v as the 4-bit code v of the fixed code-length code, most significant bit first; [257] = the one distance
       * code, of length 1 (RFC 1951 3.2.7: one distance code is spelled with one bit; no literal block ever uses it) */
      const u32 v = s <= 256u ? (u32)s_len[s] : 1u;
      bgzf_put_bits(s_out, pay0 + 74u + 4u * s, __brev(v) >> 28, 4u);
    }
    /* literals: thread t packs bytes 128 t .. 128 t + 127 */
    const u32 c0 = 128u * tid;
    const u32 cnt = c0 < n ? (n - c0 < 128u ? n - c0 : 128u) : 0u;
    const u32 full = cnt >> 2; /* whole words of the chunk; a ragged member's last bytes go one by one */
    const u32* words = s_in + 32u * tid;
    u32 my_bits = 0;
    for (u32 j = 0; j < full; j++) {
      const u32 x = words[j];
      my_bits += (s_code[x & 0xFFu] & 15u) + (s_code[(x >> 8) & 0xFFu] & 15u) + (s_code[(x >> 16) & 0xFFu] & 15u) + (s_code[x >> 24] & 15u);
    }
    for (u32 j = 4u * full; j < cnt; j++) my_bits += s_code[bytes[c0 + j]] & 15u;
    u64 total;
    const u32 ex = (u32)block_excl_scan_u64((u64)my_bits, s_w, &total);
    {
      const u32 pos = pay0 + WGA_BGZF_DYN_HDR_BITS + ex;
      u32 w = pos >> 5, nb = pos & 31u;
      u64 acc = 0;
      bool first = true;
      auto flush = [&]() { /* a full word leaves: the first one shares its word with whoever is in front */
        if (nb >= 32u) {
          if (first)
            atomicOr(&s_out[w], (u32)acc);
          else
            s_out[w] = (u32)acc;
          first = false;
          w++;
          acc >>= 32;
          nb -= 32u;
        }
      };
      auto put = [&](u32 e) {
        acc |= (u64)(e >> 4) << nb;
        nb += e & 15u;
      };
      for (u32 j = 0; j < full; j++) {
        const u32 x = words[j];
        const u32 e0 = s_code[x & 0xFFu], e1 = s_code[(x >> 8) & 0xFFu], e2 = s_code[(x >> 16) & 0xFFu], e3 = s_code[x >> 24];
        put(e0); /* two codes are at most 30 bits: 31 + 30 still fit the accumulator */
        put(e1);
        flush();
        put(e2);
        put(e3);
        flush();
      }
      for (u32 j = 4u * full; j < cnt; j++) {
        put(s_code[bytes[c0 + j]]);
        flush();
      }
      if (cnt && (nb != 0u || first)) atomicOr(&s_out[w], (u32)acc);
    }
    const u32 lit_bits = (u32)total;
    if (tid == 0u) {
      const u32 e = s_code[256];
      bgzf_put_bits(s_out, pay0 + WGA_BGZF_DYN_HDR_BITS + lit_bits, e >> 4, e & 15u);
    }
    pay_bytes = (WGA_BGZF_DYN_HDR_BITS + lit_bits + (s_code[256] & 15u) + 7u) >> 3;
  }
  if (tid == 0u) {
    const u32 t0 = pay0 + 8u * pay_bytes;
    bgzf_put_bits(s_out, t0, mb.crc, 32u);
    bgzf_put_bits(s_out, t0 + 32u, n, 32u);
  }
  __syncthreads();
  /* the image out: whole words where the member covers them, bytes at the ragged ends */
  {
    u32* gw = (u32*)(dst - a);
    const u32 end = a + msize; /* image bytes [a, end) */
    const u32 w_lo = a ? 1u : 0u, w_hi = end >> 2;
    for (u32 i = w_lo + tid; i < w_hi; i += 256u) gw[i] = s_out[i];
    const u8* img = (const u8*)s_out;
    if (a && tid >= a && tid < 4u && tid < end) dst[tid - a] = img[tid];
    const u32 tail = end & 3u;
    if (tid < tail && 4u * w_hi + tid >= a) ((u8*)gw)[4u * w_hi + tid] = img[4u * w_hi + tid];
  }
}

#endif /* WGA_K18_BGZF_DEFLATE_H */
