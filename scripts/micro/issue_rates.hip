// micro-benchmark (gfx950): instruction issue rates per CU for the integer instructions the row kernels are made of —
// how many wave-instructions per cycle a CU retires for plain 32-bit VALU (v_add_u32 / v_xor / v_bfi / v_perm), for SALU
// (s_add_u32 ...), and for both together (do scalar and vector instructions of different waves overlap?), at 4..32 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
#define REP 64

template <int MODE>
__global__ __launch_bounds__(256) void k(u32* out, int iters, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  u32 s0 = (u32)__builtin_amdgcn_readfirstlane((int)seed), s1 = s0 * 3u, s2 = s0 * 5u, s3 = s0 * 7u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
      if (MODE == 0 || MODE == 2) { /* 8 independent VALU ops */
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a1) : "v"(a2));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(a3));
        asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a3) : "v"(a4), "v"(a5));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a4) : "v"(a5));
        asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a5) : "v"(a6), "v"(a7));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a6) : "v"(a7));
        asm volatile("v_and_b32 %0, %0, %1" : "+v"(a7) : "v"(a0));
      }
      if (MODE == 1 || MODE == 2) { /* 8 SALU ops */
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
        asm volatile("s_xor_b32 %0, %0, %1" : "+s"(s1) : "s"(s2) : "scc");
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s3) : "scc");
        asm volatile("s_lshl_b32 %0, %0, 1" : "+s"(s3) : : "scc");
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s2) : "scc");
        asm volatile("s_and_b32 %0, %0, %1" : "+s"(s1) : "s"(s3) : "scc");
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s0) : "scc");
        asm volatile("s_or_b32 %0, %0, %1" : "+s"(s3) : "s"(s1) : "scc");
      }
      if (MODE == 3) { /* 8 dependent VALU ops (one chain) */
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(a2));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a3));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(a4));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a5));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(a6));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a7));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(a1));
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;
}

int main() {
  u32* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double clk = p.clockRate * 1e3; /* Hz */
  printf("CUs %d, clock %.0f MHz\n", p.multiProcessorCount, clk / 1e6);
  const int iters = 4000;
  const char* names[4] = {"VALU x8 independent", "SALU x8", "VALU x8 + SALU x8", "VALU x8 one chain"};
  for (int mode = 0; mode < 4; mode++)
    for (int bpc = 1; bpc <= 8; bpc *= 2) { /* blocks of 4 waves per CU */
      const int grid = p.multiProcessorCount * bpc;
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      auto launch = [&] {
        if (mode == 0) k<0><<<grid, 256>>>(out, iters, 1);
        if (mode == 1) k<1><<<grid, 256>>>(out, iters, 1);
        if (mode == 2) k<2><<<grid, 256>>>(out, iters, 1);
        if (mode == 3) k<3><<<grid, 256>>>(out, iters, 1);
      };
      launch();
      hipEventRecord(a);
      launch();
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double insts_per_wave = (double)iters * REP * (mode == 2 ? 2 : 1);
      const double waves_per_cu = 4.0 * bpc;
      const double per_cu_per_cycle = insts_per_wave * waves_per_cu / (ms * 1e-3 * clk);
      printf("%-22s %2.0f waves/CU: %.3f ms  -> %.3f wave-instructions / cycle / CU\n", names[mode], waves_per_cu, ms, per_cu_per_cycle);
    }
  return 0;
}
