/*
 * wga_intrin.h — the gfx950 instructions the kernels are written with, by name: wave-uniform values (readfirstlane),
 * cross-lane moves and scans (DPP, readlane, writelane, mbcnt), v_perm / v_bfe, raw buffer loads and stores with their
 * cache policies, the kernarg segment, wave barriers, the clock.  Every kernel header includes this file and nothing in
 * them is conditional on the build.
 *
 * The CPU test-suite compiles the same kernel source under a SIMT emulator (tests/emu/simt_emu.h, -DWGA_EMU): that build
 * takes the same names from tests/emu/wga_intrin_emu.h — plain C++ with the same results, test infrastructure that is
 * never part of libwgahip.so.  The emulator therefore checks the kernels' logic, not these instruction sequences; those
 * are covered by the `-m gpu` parity tests.
 */
#ifndef WGA_INTRIN_H
#define WGA_INTRIN_H

#ifdef WGA_EMU
#include "wga_intrin_emu.h"
#else

/* tell the compiler a value is wave-uniform so that it lives in SGPRs (scalar loads, no VGPRs) */
#define WGA_UNI32(x) ((u32)__builtin_amdgcn_readfirstlane((int)(x)))
#define WGA_UNI64(x)                                                               \
  (((u64)(u32)__builtin_amdgcn_readfirstlane((int)((u64)(x) >> 32)) << 32) |       \
   (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(u64)(x)))

/* index of the wave inside its block.  threadIdx.x >> 6 is the same in all 64 lanes, but the compiler's
 * divergence analysis does not know: everything derived from it (tile / record index, offsets loaded with it,
 * loop bounds) would be treated as per-lane — vector loads, exec-masked loops, VGPR-held "uniform" values.
 * WGA_WAVE_ID(t) can be switched back to the plain shift with -DWGA_WAVE_ID_PLAIN for A/B measurements. */
#ifdef WGA_WAVE_ID_PLAIN
#define WGA_WAVE_ID(t) ((u32)(t) >> 6)
#else
#define WGA_WAVE_ID(t) ((u32)__builtin_amdgcn_readfirstlane((int)((u32)(t) >> 6)))
#endif

/* keep the computation of a value where it is written (the compiler otherwise sinks LDS reads into
 * exec-masked branches "to save them", which costs more in branch overhead than the reads); the 4- and 7-value forms
 * also make several loads wait once */
#define WGA_PIN(x) asm volatile("" : "+v"(x))
#define WGA_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define WGA_PIN7(a, b, c, d, e, f, g) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g))

/* Inclusive scan of a u32 over the 64 lanes: six DPP adds (row_shr 1/2/4/8 inside each 16-lane row, then
 * row_bcast:15 and row_bcast:31 across rows): no LDS crossbar, no index arithmetic — a __shfl_up formulation costs ~5x
 * the VALU work.  Lanes that a step does not reach add the `old` operand, 0. */
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v) {
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); /* row_shr:1 */
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); /* row_shr:2 */
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); /* row_shr:4 */
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); /* row_shr:8 */
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); /* row_bcast:15 */
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); /* row_bcast:31 */
  return v;
}
/* the same with max instead of +, and the value of the lane in front (lane 0: `fill`): wave_shr:1 */
__device__ __forceinline__ u32 wave_incl_scan_max_u32(u32 v) {
  u32 t;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); v = v > t ? v : t;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); v = v > t ? v : t;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); v = v > t ? v : t;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); v = v > t ? v : t;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); v = v > t ? v : t;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); v = v > t ? v : t;
  return v;
}
__device__ __forceinline__ u32 wave_shr1_u32(u32 v, u32 fill) {
  return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false);
}
/* value of lane 63 of an inclusive scan = the wave total (uniform) */
__device__ __forceinline__ u32 wave_last_u32(u32 incl) { return (u32)__builtin_amdgcn_readlane((int)incl, 63); }
/* v with lane K's copy replaced by a wave-uniform value (v_writelane_b32 x 2; this clang has no builtin for it) */
template <u32 K>
__device__ __forceinline__ u64 lane_put_u64(u64 v, u64 uniform_val, u32 lane) {
  (void)lane;
  u32 lo = (u32)v, hi = (u32)(v >> 32);
  const u32 ulo = WGA_UNI32((u32)uniform_val), uhi = WGA_UNI32((u32)(uniform_val >> 32));
  asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"(ulo), "n"(K));
  asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"(uhi), "n"(K));
  return ((u64)hi << 32) | (u64)lo;
}

/* v_perm_b32: result byte k = byte sel[k] (0..7) of the 8-byte pool {hi:7..4, lo:3..0} */
__device__ __forceinline__ u32 byte_perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

/* Raw buffer access (128-bit descriptor in SGPRs, 32-bit per-lane byte offset).  An offset at or
 * beyond `bytes` is out of range: the load returns zeros and the store is dropped — per-lane
 * predication without exec-mask branches, which keeps the chunk loop one basic block (the
 * compiler then places its s_waitcnt exactly; with branches around the memory ops it falls back
 * to vmcnt(0) between them and serialises the loads).  Byte-unaligned offsets are fine. */
#define WGA_BUF_OOB 0xFFFFFFFFu
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef u32 u32x4_v __attribute__((vector_size(16)));
__device__ __forceinline__ BufRsrc buf_make(const void* base, u32 bytes) {
  /* the descriptor must live in SGPRs: values the compiler cannot prove wave-uniform would make it
   * wrap every access in a readfirstlane "waterfall" loop */
  const u64 b = (u64)base;
  const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)b);
  const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(b >> 32));
  const u32 n = (u32)__builtin_amdgcn_readfirstlane((int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((u64)hi << 32) | (u64)lo), (short)0, (int)n, 0x00020000);
}
__device__ __forceinline__ void buf_load16(const BufRsrc& r, u32 off, u32 v[4]) {
  const u32x4_v x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  v[0] = x[0];
  v[1] = x[1];
  v[2] = x[2];
  v[3] = x[3];
}
#ifndef WGA_STORE_AUX
#define WGA_STORE_AUX 0 /* cache policy of the row stores: 0 default, 2 nt, 16 sc1 (write-through) */
#endif
__device__ __forceinline__ void buf_store16(const BufRsrc& r, u32 off, const u32 v[4]) {
  const u32x4_v x = {v[0], v[1], v[2], v[3]};
  __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)off, 0, WGA_STORE_AUX);
}
/* streaming store (nt): for output that leaves in whole 128-byte lines and is not read again by the kernel.  On lines
 * that arrive in pieces it is much slower than the default policy (profiles/r02_k2_experiments.md). */
#ifndef WGA_STREAM_AUX
#define WGA_STREAM_AUX 2
#endif
__device__ __forceinline__ void buf_store16_stream(const BufRsrc& r, u32 off, const u32 v[4]) {
  const u32x4_v x = {v[0], v[1], v[2], v[3]};
  __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)off, 0, WGA_STREAM_AUX);
}

/* dwords k, k+1 of a value spread over the lanes of a wave (wave-uniform result) */
__device__ __forceinline__ u64 wave_get_u64(u32 v, int k) {
  return (u64)(u32)__builtin_amdgcn_readlane((int)v, k) | ((u64)(u32)__builtin_amdgcn_readlane((int)v, k + 1) << 32);
}
__device__ __forceinline__ u32 wave_get_u32(u32 v, int k) { return (u32)__builtin_amdgcn_readlane((int)v, k); }
/* v_readlane with a wave-uniform lane index */
__device__ __forceinline__ u32 wave_get_u32_dyn(u32 v, u32 k) { return (u32)__builtin_amdgcn_readlane((int)v, (int)k); }
#define WGA_CLOCK() ((u64)__builtin_amdgcn_s_memtime())
/* a polling wave steps aside for 64 * n cycles (n a constant below 128): its loads do not crowd the ones it waits for */
#define WGA_SLEEP(n) __builtin_amdgcn_s_sleep(n)
/* lanes of a wave exchange data through LDS: hardware runs them in lockstep, only the compiler must not reorder */
#define WGA_WAVE_SYNC() __builtin_amdgcn_wave_barrier()

/* set bits of m below this lane's (v_mbcnt_lo / v_mbcnt_hi; -DWGA_MBCNT=0: popcount of the masked word, for A/B) */
#ifndef WGA_MBCNT
#define WGA_MBCNT 1
#endif
__device__ __forceinline__ u32 lane_rank(u64 m, u32 lane) {
#if WGA_MBCNT
  (void)lane;
  return (u32)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
#else
  return (u32)__popcll(m & ((1ull << lane) - 1ull));
#endif
}

/* four packed ops of a tile (-DWGA_OPS_NT: as a non-temporal load, for A/B) */
__device__ __forceinline__ u32x4_a16 ops_load16(const u32* p) {
#ifdef WGA_OPS_NT
  return __builtin_nontemporal_load((const u32x4_a16*)p);
#else
  return *(const u32x4_a16*)p;
#endif
}

/* all ones when bit idx of bits is set (v_bfe_i32) */
__device__ __forceinline__ u32 bit_mask(u32 bits, u32 idx) { return (u32)__builtin_amdgcn_sbfe((int)bits, idx, 1u); }
/* ... and 1 (v_bfe_u32).  Both take the index from the low FIVE bits of idx: with a 16-bit class constant standing twice in
 * `bits` they work on a packed op as it is (bit 4, the length's lowest bit, picks one copy or the other) */
__device__ __forceinline__ u32 bit_test(u32 bits, u32 idx) { return __builtin_amdgcn_ubfe(bits, idx, 1u); }

/* kernel arguments read where they are used, from the kernarg segment (scalar loads), instead of occupying SGPRs for the
 * whole kernel: the pointer type's address space, the segment pointer, and a fence that keeps loads through it behind a point */
#define WGA_KARG_SPACE __attribute__((address_space(4)))
#define WGA_KARG_SEGMENT(T, a) ((T)__builtin_amdgcn_kernarg_segment_ptr())
#define WGA_KARG_FRESH(p) asm volatile("" : "+s"(p))

/* the window kernel's row stores: whole lines that are not read again (2 = nt; 0 default) */
#ifndef WGA_W_STORE_AUX
#define WGA_W_STORE_AUX 2
#endif
__device__ __forceinline__ void buf_store16_w(const BufRsrc& r, u32 off, const u32 v[4]) {
  const u32x4_v x = {v[0], v[1], v[2], v[3]};
  __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)off, 0, WGA_W_STORE_AUX);
}

/* ---- LDS-DMA streaming (the streaming row kernel, wga_kernels_k2s.h) ------------------------------------------------
 * lds_dma16: every active lane copies 16 bytes from gbase + voff (gbase wave-uniform, 16-byte aligned addresses) to the LDS
 * address lds_dst + 16 * lane (global_load_lds_dwordx4; the destination base travels in M0, saved and restored around the
 * instruction).  The data lands later — nothing orders an LDS read behind it but this wave's own s_waitcnt vmcnt — and the
 * instruction is INVISIBLE to the compiler's wait bookkeeping: the kernel counts its vector memory operations itself
 * (DMAs and the gstore16_nt row stores, one instruction each, in issue order: on gfx9 loads and stores retire in order on
 * one counter) and waits with vm_wait(k) = "at most k of the operations issued so far are still outstanding".  Operations
 * the compiler issues on its own (byte stores of partial granules, atomics) are not counted, which only makes a wait
 * longer than needed, never shorter. */
__device__ __forceinline__ void lds_dma16(const void* gbase, u32 voff, void* lds_dst) {
  const u64 gb = WGA_UNI64((u64)gbase);
  const u32 ld = WGA_UNI32((u32)(u64)lds_dst); /* low half of a generic LDS pointer = the LDS byte address */
  u32 keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gb), "s"(ld)
               : "memory");
}
/* the same behind this wave's own LDS reads of the destination (a slot that is refilled right after it was read) */
__device__ __forceinline__ void lds_dma16_after_reads(const void* gbase, u32 voff, void* lds_dst) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  lds_dma16(gbase, voff, lds_dst);
}
/* 16 bytes per lane to base + voff (base wave-uniform), streaming policy, one instruction the kernel counts */
__device__ __forceinline__ void gstore16_nt(void* base, u32 voff, const u32 v[4]) {
  const u64 b = WGA_UNI64((u64)base);
  const u32x4_v x = {v[0], v[1], v[2], v[3]};
  asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(voff), "v"(x), "s"(b) : "memory");
}
__device__ __forceinline__ void vm_wait(u32 k) { /* k wave-uniform */
  switch (k) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    default: asm volatile("" ::: "memory"); break; /* 16 or more may stay outstanding: nothing to wait for */
  }
}
/* v_bfi_b32: bits of a where mask is set, of b elsewhere */
__device__ __forceinline__ u32 bfi_b32(u32 mask, u32 a, u32 b) { return (a & mask) | (b & ~mask); }

#endif /* !WGA_EMU */

#endif /* WGA_INTRIN_H */
