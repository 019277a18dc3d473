// micro-benchmark: int32 atomic add rate on a 200 MB array, "difference-array marks" pattern (each wave
// walks a window: lane marks ~25 positions apart), device scope vs workgroup scope; and the XCC id a
// block runs on.  gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int SCOPE>
__global__ __launch_bounds__(256) void k(int* cov, size_t n, unsigned nwin, unsigned hot) {
  const unsigned wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  const unsigned lane = threadIdx.x & 63;
  if (wave >= nwin) return;
  // window start: pseudo-random
  size_t base = hot ? (size_t)(wave % hot) * 40000u : ((size_t)wave * 2654435761u) % (n - 64 * 16 * 25 - 64);
  for (int e = 0; e < 16; e++) {
    size_t p = base + ((size_t)(e * 64 + lane)) * 25;
    __hip_atomic_fetch_add(cov + p, 1, __ATOMIC_RELAXED, SCOPE);
    __hip_atomic_fetch_add(cov + p + 24, -1, __ATOMIC_RELAXED, SCOPE);
  }
}
__global__ void kx(unsigned* out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
  const size_t n = 50u << 20;
  int* cov;
  hipMalloc(&cov, n * 4);
  hipMemset(cov, 0, n * 4);
  const unsigned nwin = 488000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const unsigned hots[4] = {0, 0, 64, 1024};
  for (int s = 0; s < 4; s++) {
    for (int r = 0; r < 2; r++) {
      hipEventRecord(a);
      if (s == 0) k<__HIP_MEMORY_SCOPE_AGENT><<<(nwin + 3) / 4, 256>>>(cov, n, nwin, 0);
      else if (s == 1) k<__HIP_MEMORY_SCOPE_WORKGROUP><<<(nwin + 3) / 4, 256>>>(cov, n, nwin, 0);
      else k<__HIP_MEMORY_SCOPE_AGENT><<<(nwin + 3) / 4, 256>>>(cov, n, nwin, hots[s]);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      printf("%s scope (hot windows %u): %.3f ms for %.2e atomics = %.1f G atomics/s\n", s == 1 ? "workgroup" : "agent    ", hots[s], ms, nwin * 64.0 * 32, nwin * 64.0 * 32 / ms / 1e6);
    }
  }
  unsigned* xo;
  hipMalloc(&xo, 64 * 4);
  kx<<<64, 64>>>(xo);
  unsigned h[64];
  hipMemcpy(h, xo, 64 * 4, hipMemcpyDeviceToHost);
  printf("XCC_ID of blocks 0..31:");
  for (int i = 0; i < 32; i++) printf(" %u", h[i] & 0xF);
  printf("\n");
  return 0;
}
