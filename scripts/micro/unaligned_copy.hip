// micro-benchmark: 16-byte/lane copy with byte-misaligned source / destination (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4_a1 __attribute__((vector_size(16), aligned(1)));
__global__ void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t n16, int sh, int dh) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  for (; i < n16; i += stride) {
    u32x4_a1 v = *(const u32x4_a1*)(src + 16 * i + sh);
    *(u32x4_a1*)(dst + 16 * i + dh) = v;
  }
}
int main() {
  size_t bytes = (size_t)2 << 30;
  unsigned char *s, *d;
  hipMalloc(&s, bytes + 64); hipMalloc(&d, bytes + 64);
  hipMemset(s, 1, bytes + 64); hipMemset(d, 0, bytes + 64);
  int cfg[6][2] = {{0,0},{3,0},{0,3},{3,5},{1,1},{8,8}};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (auto& c : cfg) {
    k<<<256 * 16, 256>>>(s, d, bytes / 16, c[0], c[1]);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k<<<256 * 16, 256>>>(s, d, bytes / 16, c[0], c[1]);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("src+%d dst+%d : %.1f GB/s (read+write)\n", c[0], c[1], 2.0 * bytes * 5 / (ms * 1e-3) / 1e9);
  }
  return 0;
}
