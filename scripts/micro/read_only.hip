// micro-benchmark: what does a kernel that only READS reach on this part?  K1 (stat), K5's list pass, K7 / K12's count passes
// all stream the packed ops once and write next to nothing, and all of them sit at 3.5-3.7 TB/s whatever their instruction
// count or the number of loads they keep in flight.  A sum over a buffer, 16 B per lane and load:
//   U loads in flight per thread (1 / 4 / 8), a grid of one block per 256 x U x 16 bytes or a grid of resident blocks that
//   stride over the buffer; buffers of 2 GB (configs[1]'s ops) and 32 GB (far beyond the 256 MB Infinity Cache);
//   and, for comparison, a copy (read + write) of the same bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((vector_size(16)));

template <int U, bool PERSIST>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ src, size_t n16, unsigned* out) {
  size_t i = ((size_t)blockIdx.x * 256 * U) + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  u32x4 acc = {0, 0, 0, 0};
  do {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = i + (size_t)u * 256 < n16 ? src[i + (size_t)u * 256] : acc;
#pragma unroll
    for (int u = 0; u < U; u++) acc ^= v[u];
    i += stride;
  } while (PERSIST && i < n16);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;  // keeps the loads alive, almost never true
}
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  u32x4 v[4];
#pragma unroll
  for (int u = 0; u < 4; u++) v[u] = i + (size_t)u * 256 < n16 ? src[i + (size_t)u * 256] : v[0];
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (i + (size_t)u * 256 < n16) dst[i + (size_t)u * 256] = v[u];
}
template <typename F>
static double timed(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipEventRecord(a);
  for (int r = 0; r < reps; r++) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}
int main() {
  unsigned* out;
  hipMalloc(&out, 64);
  for (size_t gb : {2ull, 32ull}) {
    const size_t bytes = gb << 30, n16 = bytes / 16;
    u32x4 *s, *d;
    if (hipMalloc(&s, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess) { printf("no memory for %zu GB\n", gb); return 1; }
    hipMemset(s, 0x5A, bytes);
    hipMemset(d, 0, bytes);
    const int reps = gb == 2 ? 20 : 4;
    double ms;
    ms = timed([&] { k_read<1, false><<<(unsigned)((n16 + 255) / 256), 256>>>(s, n16, out); }, reps);
    printf("%2zu GB read  1 load / thread, block per KB-4   : %7.3f ms  %6.0f GB/s\n", gb, ms, bytes / ms / 1e6);
    ms = timed([&] { k_read<4, false><<<(unsigned)((n16 + 1023) / 1024), 256>>>(s, n16, out); }, reps);
    printf("%2zu GB read  4 loads / thread, block per 16 KB  : %7.3f ms  %6.0f GB/s\n", gb, ms, bytes / ms / 1e6);
    ms = timed([&] { k_read<8, false><<<(unsigned)((n16 + 2047) / 2048), 256>>>(s, n16, out); }, reps);
    printf("%2zu GB read  8 loads / thread, block per 32 KB  : %7.3f ms  %6.0f GB/s\n", gb, ms, bytes / ms / 1e6);
    for (int per_cu : {4, 8, 16}) {
      ms = timed([&] { k_read<4, true><<<256 * per_cu, 256>>>(s, n16, out); }, reps);
      printf("%2zu GB read  4 loads / thread, %2d resident blocks per CU striding : %7.3f ms  %6.0f GB/s\n", gb, per_cu, ms, bytes / ms / 1e6);
    }
    ms = timed([&] { k_copy<<<(unsigned)((n16 + 1023) / 1024), 256>>>(s, d, n16); }, reps);
    printf("%2zu GB copy  4 x 16 B / thread                  : %7.3f ms  %6.0f GB/s read + as much written = %6.0f GB/s\n", gb, ms, bytes / ms / 1e6, 2.0 * bytes / ms / 1e6);
    hipFree(s);
    hipFree(d);
  }
  return 0;
}
