/*
 * wga_k15_fasta.h — K15: FASTA text in HBM -> line-stripped sequence pool + contig table.
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K15_FASTA_H
#define WGA_K15_FASTA_H

#include "wga_kernels.h"

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

#endif /* WGA_K15_FASTA_H */
