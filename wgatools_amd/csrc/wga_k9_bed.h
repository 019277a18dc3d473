/*
 * wga_k9_bed.h — K9: pafcov BED text (pafcov.rs:56-60).
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K9_BED_H
#define WGA_K9_BED_H

#include "wga_kernels.h"

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

/* one BED line at p (LDS or memory) */
template <typename P>
__device__ __forceinline__ void bed_line(P p, const ScanCovLine& f, const u8* __restrict__ name, u32 i) {
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
/* A block takes WGA_BED_LINES consecutive lines — one contiguous stretch of the text.  Its threads put their lines into an LDS
 * buffer that mirrors the stretch's position inside its 16-byte group and the stretch leaves in 16-byte stores (lds_text_flush):
 * a thread per line writing its ~28 bytes one by one to memory was 0.53 TB/s of text (rounds 1-5).  A stretch longer than the
 * buffer (names of more than ~30 bytes) is written directly as before. */
#define WGA_BED_LINES 512u
#define WGA_BED_STAGE 24576u
__global__ __launch_bounds__(256) void k_pafcov_format(ScanCovLine f, u32 n, const u8* __restrict__ name,
                                                       const u64* __restrict__ line_off,
                                                       u8* __restrict__ out) {
  __shared__ u32x4_a16 s_buf[(WGA_BED_STAGE + 32u) / 16u];
  const u32 tid = threadIdx.x;
  const u32 x0 = blockIdx.x * WGA_BED_LINES, x1 = x0 + WGA_BED_LINES < n ? x0 + WGA_BED_LINES : n;
  const u64 first = line_off[0], e0 = line_off[x0], e1 = line_off[x1];
  u8* const g0 = out + (e0 - first);
  if (e1 - e0 <= (u64)WGA_BED_STAGE) { /* block-uniform */
    const u32 a = (u32)((uintptr_t)g0 & 15u);
    u8* const tbuf = (u8*)s_buf;
    for (u32 x = x0 + tid; x < x1; x += 256u) bed_line(tbuf + a + (u32)(line_off[x] - e0), f, name, x);
    __syncthreads();
    lds_text_flush(tbuf, a, (u32)(e1 - e0), g0 - a, tid, 256u);
    return;
  }
  for (u32 x = x0 + tid; x < x1; x += 256u) bed_line(out + (line_off[x] - first), f, name, x);
}

#endif /* WGA_K9_BED_H */
