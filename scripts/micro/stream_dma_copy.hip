// micro-benchmark (gfx950): can one wave stream a row of K2's shape — source bytes staged through an LDS ring by
// LDS-DMA (global_load_lds_dwordx4, several chunks in flight, counted vmcnt), read back byte-unaligned, stored as whole
// 16-byte-aligned 1 KB steps — at the rate of a plain copy?  No CIGAR logic: output column c reads source byte
// p(c) = c - GAP * (c / PERIOD), i.e. the source offset slips by GAP bytes every PERIOD columns as it does at a gap.
//   direct    unaligned 16 B global loads (U steps in flight), aligned 16 B stores            (v1's fast path)
//   ring      R x 1 KB LDS ring per wave, D chunks ahead, ds_read_b128 at the byte offset, aligned nt stores
//   ring2     ... two windows per granule merged under a byte mask (what a gap inside a granule costs in LDS reads)
// Parameters of the ring kernels: waves per block, ring slots R, depth D (LDS per wave = R KB + 16 B).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned int u32x4_a1 __attribute__((vector_size(16), aligned(1)));
typedef unsigned int u32x4_a16 __attribute__((vector_size(16), aligned(16)));
#define JOB_BYTES 65536u /* output bytes of one wave's job */
#define PERIOD 151u
#define GAP 3u

__device__ __forceinline__ u32 p_of(u32 c) { return c - GAP * (c / PERIOD); }

__device__ __forceinline__ void dma16(const unsigned char* gsrc, u32 lds_dst) {
  u32 keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void store16_nt(unsigned char* p, u32x4_a16 v) {
  asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void wait_vm(u32 k) { /* k wave-uniform: at most k vector memory operations outstanding */
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
    default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
  }
}

template <int U>
__global__ __launch_bounds__(256) void k_direct(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                size_t njobs, size_t src_span) {
  const u32 lane = threadIdx.x & 63u;
  const size_t j = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= njobs) return;
  const unsigned char* s = src + (j * (size_t)JOB_BYTES) % src_span + (j * 7u) % 16u;
  unsigned char* d = dst + j * (size_t)JOB_BYTES;
  for (u32 st = 0; st < JOB_BYTES / 1024u; st += U) {
    u32x4_a1 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = *(const u32x4_a1*)(s + p_of((st + u) * 1024u + lane * 16u));
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store((u32x4_a16)v[u], (u32x4_a16*)(d + (st + u) * 1024u + lane * 16u));
  }
}

template <int WAVES, int R, int D, int MERGE>
__global__ __launch_bounds__(WAVES * 64) void k_ring(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                     size_t njobs, size_t src_span) {
  __shared__ __attribute__((aligned(16))) unsigned char s_ring[WAVES * (R * 1024 + 16)];
  const u32 lane = threadIdx.x & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t j = (size_t)blockIdx.x * WAVES + wave;
  if (j >= njobs) return;
  const unsigned char* s = src + (j * (size_t)JOB_BYTES) % src_span + (j * 7u) % 16u;
  unsigned char* d = dst + j * (size_t)JOB_BYTES;
  const u32 a_off = (u32)((size_t)s & 15u);
  const unsigned char* A = s - a_off; /* 16-byte aligned stream base: chunk k = A + 1024 k */
  unsigned char* ring = s_ring + wave * (R * 1024 + 16);
  const u32 ring_lds = (u32)(size_t)ring; /* LDS byte address (low 32 bits of the shared pointer) */
  const u32 src_bytes = p_of(JOB_BYTES - 1u) + 1u + 16u;
  const u32 nchunks = (src_bytes + a_off + 1023u) >> 10;
  u32 issued = 0;   /* chunks [0, issued) have their DMA issued */
  u32 vm = 0;       /* vector memory operations issued by this wave (the ones counted: DMAs and step stores) */
  u64 vm_at = 0;    /* byte (k & 7): vm right after chunk k's DMA */
  for (u32 st = 0; st < JOB_BYTES / 1024u; st++) {
    const u32 c0 = st * 1024u + lane * 16u;
    const u32 p_end = p_of(st * 1024u + 1023u) + a_off + 16u;
    u32 kn = p_end >> 10;
    if (kn >= nchunks) kn = nchunks - 1u;
    u32 want = kn + (u32)D + 1u;
    if (want > nchunks) want = nchunks;
    while (issued < want) {
      const u32 slot = issued % (u32)R;
      dma16(A + (size_t)issued * 1024u + lane * 16u, ring_lds + slot * 1024u);
      vm++;
      if (slot == 0u) { /* the ring's first 16 bytes again behind its end: windows that start in the last slot read across */
        if (lane == 0u) dma16(A + (size_t)issued * 1024u, ring_lds + (u32)R * 1024u);
        vm++;
      }
      vm_at = (vm_at & ~(0xFFull << (8u * (issued & 7u)))) | ((u64)(vm & 0xFFu) << (8u * (issued & 7u)));
      issued++;
    }
    const u32 after = (vm - (u32)(vm_at >> (8u * (kn & 7u)))) & 0xFFu; /* operations issued behind chunk kn's DMA */
    wait_vm(after);
    const u32 p0 = p_of(c0) + a_off;
    const u32 pos0 = p0 % ((u32)R * 1024u);
    u32x4_a1 w = *(const u32x4_a1*)(ring + pos0);
    if (MERGE) {
      const u32 p1 = p_of(c0 + 15u) + a_off - 15u;
      const u32 pos1 = p1 % ((u32)R * 1024u);
      const u32x4_a1 w1 = *(const u32x4_a1*)(ring + pos1);
      const u32 cut = PERIOD - (c0 % PERIOD); /* bytes [cut, 16) come from the second window */
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int lo = (int)cut - 4 * q;
        const u32 m = lo >= 4 ? 0xFFFFFFFFu : (lo <= 0 ? 0u : ((1u << (8 * lo)) - 1u));
        w[q] = (w[q] & m) | (w1[q] & ~m);
      }
    }
    store16_nt(d + c0, (u32x4_a16)w);
    vm++;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static float time_it(void (*launch)(void*), void* ctx, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch(ctx);
  launch(ctx);
  hipEventRecord(a);
  for (int r = 0; r < reps; r++) launch(ctx);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a);
  hipEventDestroy(b);
  return ms / reps;
}

struct Ctx {
  const unsigned char* s;
  unsigned char* d;
  size_t njobs, span;
};
template <int U>
static void l_direct(void* p) {
  Ctx* c = (Ctx*)p;
  k_direct<U><<<(unsigned)((c->njobs + 3) / 4), 256>>>(c->s, c->d, c->njobs, c->span);
}
template <int WAVES, int R, int D, int MERGE>
static void l_ring(void* p) {
  Ctx* c = (Ctx*)p;
  k_ring<WAVES, R, D, MERGE><<<(unsigned)((c->njobs + WAVES - 1) / WAVES), WAVES * 64>>>(c->s, c->d, c->njobs, c->span);
}

static int check(const Ctx& c, const unsigned char* h_src, size_t nsample, int merged) {
  /* compare a few jobs with the definition */
  unsigned char* h = (unsigned char*)malloc(JOB_BYTES);
  int bad = 0;
  for (size_t k = 0; k < nsample; k++) {
    const size_t j = (k * 7919u) % c.njobs;
    hipMemcpy(h, c.d + j * (size_t)JOB_BYTES, JOB_BYTES, hipMemcpyDeviceToHost);
    const size_t so = (j * (size_t)JOB_BYTES) % c.span + (j * 7u) % 16u;
    for (u32 col = 0; col < JOB_BYTES; col++) {
      const u32 cg = merged ? col : (col & ~15u); /* without the merge a granule is 16 bytes from its first column's place */
      const u32 p = cg - GAP * (cg / PERIOD) + (col - cg);
      if (h[col] != h_src[so + p]) {
        if (bad < 5) printf("  mismatch job %zu col %u: %u != %u\n", j, col, h[col], h_src[so + p]);
        bad++;
      }
    }
  }
  free(h);
  return bad;
}

int main(int argc, char** argv) {
  const size_t out_bytes = (size_t)6 << 30;
  const size_t njobs = out_bytes / JOB_BYTES;
  const char* only = argc > 1 ? argv[1] : "";
  for (int big = 0; big < 2; big++) {
    const size_t src_bytes = big ? ((size_t)4 << 30) : ((size_t)96 << 20);
    unsigned char *s, *d;
    if (hipMalloc(&s, src_bytes + (1 << 20)) != hipSuccess || hipMalloc(&d, out_bytes + 4096) != hipSuccess) {
      printf("alloc failed\n");
      return 1;
    }
    unsigned char* h_src = (unsigned char*)malloc(src_bytes + (1 << 20));
    u64 x = 88172645463325252ull;
    for (size_t i = 0; i < src_bytes + (1 << 20); i += 8) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      memcpy(h_src + i, &x, 8);
    }
    hipMemcpy(s, h_src, src_bytes + (1 << 20), hipMemcpyHostToDevice);
    hipMemset(d, 0, out_bytes + 4096);
    Ctx c{s, d, njobs, src_bytes - JOB_BYTES};
    printf("== source %s (%zu MB), output 6 GB, %zu jobs of 64 KB\n", big ? "in HBM" : "cache-resident", src_bytes >> 20, njobs);
#define RUN(name, fn, chk)                                                                              \
  if (!only[0] || strstr(name, only)) {                                                                 \
    hipMemset(d, 0, out_bytes);                                                                         \
    const float ms = time_it(fn, &c, 5);                                                                \
    int bad = chk ? check(c, h_src, 8, chk == 2) : -1;                                                            \
    printf("%-36s %8.3f ms  %7.1f GB/s written  (check: %d bad)\n", name, ms, out_bytes / (ms * 1e-3) / 1e9, bad); \
    fflush(stdout);                                                                                     \
  }
    RUN("direct U=1", (l_direct<1>), 1);
    RUN("direct U=4", (l_direct<4>), 1);
    RUN("ring W4 R6 D3", (l_ring<4, 6, 3, 0>), 1);
    RUN("ring W4 R8 D5", (l_ring<4, 8, 5, 0>), 1);
    RUN("ring W4 R10 D6", (l_ring<4, 10, 6, 0>), 1);
    RUN("ring W4 R4 D1", (l_ring<4, 4, 1, 0>), 1);
    RUN("ring W2 R6 D3", (l_ring<2, 6, 3, 0>), 1);
    RUN("ring W1 R6 D3", (l_ring<1, 6, 3, 0>), 1);
    RUN("ring W1 R8 D5", (l_ring<1, 8, 5, 0>), 1);
    RUN("ring2 W4 R6 D3", (l_ring<4, 6, 3, 1>), 2);
    RUN("ring2 W4 R8 D5", (l_ring<4, 8, 5, 1>), 2);
    RUN("ring2 W2 R8 D5", (l_ring<2, 8, 5, 1>), 2);
    hipFree(s);
    hipFree(d);
    free(h_src);
  }
  return 0;
}
