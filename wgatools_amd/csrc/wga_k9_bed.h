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

#endif /* WGA_K9_BED_H */
