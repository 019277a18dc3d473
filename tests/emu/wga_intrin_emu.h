/*
 * wga_intrin_emu.h — TEST INFRASTRUCTURE: the names of wgatools_amd/csrc/wga_intrin.h for the SIMT-emulator build of the
 * kernels (tests/emu/libwgaemu.so, -DWGA_EMU).  Plain C++ on the emulator's lanes (fibers) with the results the gfx950
 * instructions give; never compiled into libwgahip.so.
 */
#ifndef WGA_INTRIN_EMU_H
#define WGA_INTRIN_EMU_H

#define WGA_UNI32(x) ((u32)(x))
#define WGA_UNI64(x) ((u64)(x))

#define WGA_WAVE_ID(t) ((u32)(t) >> 6)

#define WGA_PIN(x) ((void)0)
#define WGA_PIN4(a, b, c, d) ((void)0)
#define WGA_PIN7(a, b, c, d, e, f, g) ((void)0)

__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v) {
  const u32 lane = threadIdx.x & 63u;
  for (u32 d = 1; d < 64; d <<= 1) {
    u32 t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ u32 wave_incl_scan_max_u32(u32 v) {
  const u32 lane = threadIdx.x & 63u;
  for (u32 d = 1; d < 64; d <<= 1) {
    u32 t = __shfl_up(v, d);
    if (lane >= d && t > v) v = t;
  }
  return v;
}
__device__ __forceinline__ u32 wave_shr1_u32(u32 v, u32 fill) {
  const u32 t = __shfl_up(v, 1u);
  return (threadIdx.x & 63u) == 0u ? fill : t;
}
__device__ __forceinline__ u32 wave_last_u32(u32 incl) { return __shfl(incl, 63); }
template <u32 K>
__device__ __forceinline__ u64 lane_put_u64(u64 v, u64 uniform_val, u32 lane) {
  return lane == K ? uniform_val : v;
}

__device__ __forceinline__ u32 byte_perm(u32 hi, u32 lo, u32 sel) {
  u64 pool = ((u64)hi << 32) | (u64)lo;
  u32 r = 0;
  for (int k = 0; k < 4; k++) r |= (u32)((pool >> (8 * ((sel >> (8 * k)) & 7u))) & 0xFFu) << (8 * k);
  return r;
}

#define WGA_BUF_OOB 0xFFFFFFFFu
struct BufRsrc {
  u8* base;
  u32 bytes;
};
__device__ __forceinline__ BufRsrc buf_make(const void* base, u32 bytes) {
  BufRsrc r;
  r.base = (u8*)base;
  r.bytes = bytes;
  return r;
}
__device__ __forceinline__ void buf_load16(const BufRsrc& r, u32 off, u32 v[4]) {
  for (int d = 0; d < 4; d++) {
    v[d] = 0u;
    if ((u64)off + 4u * d + 4u <= (u64)r.bytes) memcpy(&v[d], r.base + off + 4 * d, 4);
  }
}
__device__ __forceinline__ void buf_store16(const BufRsrc& r, u32 off, const u32 v[4]) {
  for (int d = 0; d < 4; d++)
    if ((u64)off + 4u * d + 4u <= (u64)r.bytes) memcpy(r.base + off + 4 * d, &v[d], 4);
}
__device__ __forceinline__ void buf_store16_stream(const BufRsrc& r, u32 off, const u32 v[4]) { buf_store16(r, off, v); }

__device__ __forceinline__ u64 wave_get_u64(u32 v, int k) { return (u64)__shfl(v, k) | ((u64)__shfl(v, k + 1) << 32); }
__device__ __forceinline__ u32 wave_get_u32(u32 v, int k) { return __shfl(v, k); }
__device__ __forceinline__ u32 wave_get_u32_dyn(u32 v, u32 k) { return __shfl(v, (int)k); }
#define WGA_CLOCK() 0ull
#define WGA_SLEEP(n) ((void)0)
/* the emulator's lanes are fibers: a real wave barrier */
#define WGA_WAVE_SYNC() emu::barrier_wait(emu::S().wave_bar[emu::flat_tid() >> 6])

__device__ __forceinline__ u32 lane_rank(u64 m, u32 lane) { return (u32)__popcll(m & ((1ull << lane) - 1ull)); }

__device__ __forceinline__ u32x4_a16 ops_load16(const u32* p) { return *(const u32x4_a16*)p; }

__device__ __forceinline__ u32 bit_mask(u32 bits, u32 idx) { return 0u - ((bits >> (idx & 31u)) & 1u); }
__device__ __forceinline__ u32 bit_test(u32 bits, u32 idx) { return (bits >> (idx & 31u)) & 1u; }

#define WGA_KARG_SPACE
#define WGA_KARG_SEGMENT(T, a) (&(a))
#define WGA_KARG_FRESH(p) ((void)0)

__device__ __forceinline__ void buf_store16_w(const BufRsrc& r, u32 off, const u32 v[4]) { buf_store16(r, off, v); }

/* LDS-DMA of the streaming row kernel: the copy happens at once (the emulator cannot show a read that comes too early —
 * the `-m gpu` parity tests do), the waits are empty */
__device__ __forceinline__ void lds_dma16(const void* gbase, u32 voff, void* lds_dst) {
  memcpy((u8*)lds_dst + 16u * (threadIdx.x & 63u), (const u8*)gbase + voff, 16);
}
__device__ __forceinline__ void lds_dma16_after_reads(const void* gbase, u32 voff, void* lds_dst) { lds_dma16(gbase, voff, lds_dst); }
__device__ __forceinline__ void gstore16_nt(void* base, u32 voff, const u32 v[4]) { memcpy((u8*)base + voff, v, 16); }
__device__ __forceinline__ void vm_wait(u32) {}
__device__ __forceinline__ u32 bfi_b32(u32 mask, u32 a, u32 b) { return (a & mask) | (b & ~mask); }

#endif /* WGA_INTRIN_EMU_H */
