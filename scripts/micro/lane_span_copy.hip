// micro-benchmark: how much does HBM throughput suffer when every lane copies SPAN contiguous
// bytes (SPAN/16 byte-misaligned 16 B load+store pairs, lane stride SPAN) instead of the
// wave-coalesced pattern (lane stride 16)?  gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4_a1 __attribute__((vector_size(16), aligned(1)));

template <int SPAN>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                          size_t nspan, int sh, int dh) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  for (; i < nspan; i += stride) {
    const unsigned char* s = src + (size_t)SPAN * i + sh;
    unsigned char* d = dst + (size_t)SPAN * i + dh;
    u32x4_a1 v[SPAN / 16];
#pragma unroll
    for (int j = 0; j < SPAN / 16; j++) v[j] = *(const u32x4_a1*)(s + 16 * j);
#pragma unroll
    for (int j = 0; j < SPAN / 16; j++) *(u32x4_a1*)(d + 16 * j) = v[j];
  }
}
// same span per lane but one 16 B pair at a time (what a per-lane piece loop does)
template <int SPAN>
__global__ __launch_bounds__(256) void kseq(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                             size_t nspan, int sh, int dh) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  for (; i < nspan; i += stride) {
    const unsigned char* s = src + (size_t)SPAN * i + sh;
    unsigned char* d = dst + (size_t)SPAN * i + dh;
#pragma nounroll
    for (int j = 0; j < SPAN / 16; j++) {
      u32x4_a1 v = *(const u32x4_a1*)(s + 16 * j);
      *(u32x4_a1*)(d + 16 * j) = v;
    }
  }
}
template <typename F>
void run(const char* name, F launch, size_t bytes) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch();
  hipEventRecord(a);
  for (int r = 0; r < 5; r++) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("%-28s %.1f GB/s (read+write)\n", name, 2.0 * bytes * 5 / (ms * 1e-3) / 1e9);
}
int main() {
  size_t bytes = (size_t)4 << 30;
  unsigned char *s, *d;
  hipMalloc(&s, bytes + 256);
  hipMalloc(&d, bytes + 256);
  hipMemset(s, 1, bytes + 256);
  hipMemset(d, 0, bytes + 256);
  const int G = 256 * 32;
  for (int sh = 0; sh <= 3; sh += 3) {
    printf("-- src+%d dst+%d\n", sh, sh ? 5 : 0);
    int dh = sh ? 5 : 0;
    run("span16 (coalesced)", [&] { k<16><<<G, 256>>>(s, d, bytes / 16, sh, dh); }, bytes);
    run("span32 batched", [&] { k<32><<<G, 256>>>(s, d, bytes / 32, sh, dh); }, bytes);
    run("span64 batched", [&] { k<64><<<G, 256>>>(s, d, bytes / 64, sh, dh); }, bytes);
    run("span128 batched", [&] { k<128><<<G, 256>>>(s, d, bytes / 128, sh, dh); }, bytes);
    run("span32 sequential", [&] { kseq<32><<<G, 256>>>(s, d, bytes / 32, sh, dh); }, bytes);
    run("span64 sequential", [&] { kseq<64><<<G, 256>>>(s, d, bytes / 64, sh, dh); }, bytes);
    run("span128 sequential", [&] { kseq<128><<<G, 256>>>(s, d, bytes / 128, sh, dh); }, bytes);
    run("span256 sequential", [&] { kseq<256><<<G, 256>>>(s, d, bytes / 256, sh, dh); }, bytes);
  }
  return 0;
}
