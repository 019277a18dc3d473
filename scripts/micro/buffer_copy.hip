// micro-check: raw buffer b128 loads/stores with byte-misaligned offsets and out-of-range
// predication (offset beyond num_records => load returns 0, store is dropped) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((vector_size(16)));

__global__ __launch_bounds__(256) void k(const unsigned char* src, unsigned char* dst, unsigned n16, unsigned bytes,
                                          int sh, int dh, int skip_mod) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + sh), (short)0, (int)bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + dh), (short)0, (int)bytes, 0x00020000);
  unsigned i = blockIdx.x * 256 + threadIdx.x;
  unsigned stride = gridDim.x * 256;
  for (; i < n16; i += stride) {
    const bool on = skip_mod == 0 || (i % (unsigned)skip_mod) != 0;
    const unsigned off = on ? 16u * i : 0xFFFFFFF0u;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)off, 0, 0);
  }
}
int main() {
  const unsigned bytes = 1u << 30;
  unsigned char *s, *d;
  hipMalloc(&s, (size_t)bytes + 64);
  hipMalloc(&d, (size_t)bytes + 64);
  std::vector<unsigned char> h(1 << 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 131 + (i >> 8));
  for (size_t o = 0; o < (size_t)bytes + 64; o += h.size()) hipMemcpy(s + o, h.data(), o + h.size() <= (size_t)bytes + 64 ? h.size() : (size_t)bytes + 64 - o, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  int cfg[4][3] = {{0, 0, 0}, {3, 5, 0}, {3, 5, 7}, {1, 2, 3}};
  for (auto& c : cfg) {
    hipMemset(d, 0xEE, (size_t)bytes + 64);
    k<<<256 * 16, 256>>>(s, d, bytes / 16, bytes, c[0], c[1], c[2]);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k<<<256 * 16, 256>>>(s, d, bytes / 16, bytes, c[0], c[1], c[2]);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // verify the first MiB
    std::vector<unsigned char> out(h.size() + 64);
    hipMemcpy(out.data(), d, out.size(), hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < h.size() / 16; i++) {
      const bool on = c[2] == 0 || (i % (unsigned)c[2]) != 0;
      for (int j = 0; j < 16; j++) {
        unsigned char want = on ? h[(16 * i + j + c[0]) % h.size()] : 0xEE;
        if (out[16 * i + j + c[1]] != want) bad++;
      }
    }
    printf("src+%d dst+%d skip every %d : %.1f GB/s (read+write, nominal)  mismatching bytes in first MiB: %zu\n", c[0], c[1], c[2],
           2.0 * bytes * 5 / (ms * 1e-3) / 1e9, bad);
  }
  return 0;
}
