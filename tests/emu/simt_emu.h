/*
 * simt_emu.h — a tiny single-threaded SIMT emulator.  TEST INFRASTRUCTURE ONLY.
 *
 * The HIP kernels in wgatools_amd/csrc/wga_kernels.h are written against a small surface
 * (threadIdx/blockIdx, __shared__, __syncthreads, __shfl*, __ballot, atomics).  This header
 * provides that surface for a plain g++ build so the *same kernel source* can be executed on the
 * CPU by the `-m "not gpu"` tests (tests/emu/libwgaemu.so) and compared with the oracle before a
 * GPU is available.  It is never linked into libwgahip.so and the product never loads it.
 *
 * Model: one block at a time; every thread of the block is a ucontext fiber; barriers (block
 * level for __syncthreads, wave level for the 64-lane cross-lane ops) yield to a round-robin
 * scheduler.  A full scheduler round without progress means the kernel has divergent barriers /
 * cross-lane ops — reported as a fatal error, which is a useful lint for real-GPU bugs.
 */
#ifndef WGA_SIMT_EMU_H
#define WGA_SIMT_EMU_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <mutex>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 {
  unsigned x, y, z, w;
};
struct uint2 {
  unsigned x, y;
};
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) {
  uint4 r = {x, y, z, w};
  return r;
}

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace emu {

struct Barrier {
  int n = 0, count = 0;
  unsigned gen = 0;
};

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
};

struct State {
  dim3 gridDim, blockDim, blockIdx, threadIdx;
  std::vector<Fiber> fibers;
  int cur = -1;
  ucontext_t sched;
  Barrier block_bar;
  Barrier wave_bar[32];
  uint64_t xchg[2048];
  std::function<void()> body;
  unsigned long progress = 0;
};

inline State& S() {
  static State s;
  return s;
}

inline void yield() {
  State& s = S();
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

inline void barrier_wait(Barrier& b) {
  State& s = S();
  unsigned gen = b.gen;
  if (++b.count == b.n) {
    b.count = 0;
    b.gen++;
    s.progress++;
  } else {
    while (b.gen == gen) yield();
  }
}

inline void trampoline() {
  State& s = S();
  s.body();
  s.fibers[s.cur].done = true;
  s.progress++;
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

static const size_t kStack = 256 * 1024;

/* one kernel at a time: the emulator has ONE set of fibers and `__shared__` is function-static storage, while the host
 * layer may drive several contexts ("devices", WGA_EMU_DEVICES) from several threads */
inline std::mutex& launch_mutex() {
  static std::mutex m;
  return m;
}

inline void launch(dim3 grid, dim3 block, std::function<void()> body) {
  std::lock_guard<std::mutex> lock(launch_mutex());
  State& s = S();
  unsigned nt = block.x * block.y * block.z;
  if (nt > 1024 || nt % 64 != 0) {
    fprintf(stderr, "emu: block size %u unsupported\n", nt);
    abort();
  }
  s.gridDim = grid;
  s.blockDim = block;
  s.body = body;
  if (s.fibers.size() < nt) {
    size_t old = s.fibers.size();
    s.fibers.resize(nt);
    for (size_t i = old; i < nt; i++) s.fibers[i].stack = (char*)malloc(kStack);
  }
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        s.blockIdx = dim3(bx, by, bz);
        s.block_bar = Barrier();
        s.block_bar.n = (int)nt;
        for (unsigned w = 0; w < nt / 64; w++) {
          s.wave_bar[w] = Barrier();
          s.wave_bar[w].n = 64;
        }
        for (unsigned t = 0; t < nt; t++) {
          Fiber& f = s.fibers[t];
          f.done = false;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = kStack;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, (void (*)())trampoline, 0);
        }
        unsigned remaining = nt;
        while (remaining) {
          unsigned long before = s.progress;
          remaining = 0;
          for (unsigned t = 0; t < nt; t++) {
            if (s.fibers[t].done) continue;
            s.cur = (int)t;
            s.threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            swapcontext(&s.sched, &s.fibers[t].ctx);
            if (!s.fibers[t].done) remaining++;
          }
          if (remaining && s.progress == before) {
            fprintf(stderr,
                    "emu: deadlock in block (%u,%u,%u): divergent barrier or cross-lane op\n", bx,
                    by, bz);
            abort();
          }
        }
      }
}

inline unsigned flat_tid() {
  State& s = S();
  return s.threadIdx.x + s.blockDim.x * (s.threadIdx.y + s.blockDim.y * s.threadIdx.z);
}

/* cross-lane exchange inside one 64-wide wave */
template <typename T>
inline T xchg(T v, unsigned src_lane, bool src_valid) {
  static_assert(sizeof(T) <= 8, "xchg");
  State& s = S();
  unsigned tid = flat_tid();
  unsigned wave = tid >> 6, lane = tid & 63;
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  s.xchg[wave * 64 + lane] = bits;
  barrier_wait(s.wave_bar[wave]);
  uint64_t got = src_valid ? s.xchg[wave * 64 + (src_lane & 63)] : bits;
  barrier_wait(s.wave_bar[wave]);
  T r;
  memcpy(&r, &got, sizeof(T));
  return r;
}

}  // namespace emu

#define threadIdx (emu::S().threadIdx)
#define blockIdx (emu::S().blockIdx)
#define blockDim (emu::S().blockDim)
#define gridDim (emu::S().gridDim)

static inline void __syncthreads() { emu::barrier_wait(emu::S().block_bar); }

template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
  (void)width;
  return emu::xchg(v, (unsigned)src, true);
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  unsigned lane = emu::flat_tid() & 63;
  return emu::xchg(v, lane ^ (unsigned)mask, true);
}
template <typename T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
  (void)width;
  unsigned lane = emu::flat_tid() & 63;
  return emu::xchg(v, lane - delta, lane >= delta);
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  (void)width;
  unsigned lane = emu::flat_tid() & 63;
  return emu::xchg(v, lane + delta, lane + delta < 64);
}
static inline unsigned long long __ballot(int pred) {
  emu::State& s = emu::S();
  unsigned tid = emu::flat_tid();
  unsigned wave = tid >> 6, lane = tid & 63;
  s.xchg[wave * 64 + lane] = pred ? 1 : 0;
  emu::barrier_wait(s.wave_bar[wave]);
  unsigned long long m = 0;
  for (unsigned l = 0; l < 64; l++)
    if (s.xchg[wave * 64 + l]) m |= 1ull << l;
  emu::barrier_wait(s.wave_bar[wave]);
  return m;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }

template <typename T>
static inline T atomicAdd(T* p, T v) {
  T old = *p;
  *p = old + v;
  return old;
}
template <typename T>
static inline T atomicOr(T* p, T v) {
  T old = *p;
  *p = old | v;
  return old;
}
static inline unsigned __brev(unsigned v) {
  unsigned r = 0;
  for (int k = 0; k < 32; k++) r |= ((v >> k) & 1u) << (31 - k);
  return r;
}
template <typename T>
static inline T atomicMin(T* p, T v) {
  T old = *p;
  if (v < old) *p = v;
  return old;
}
template <typename T>
static inline T atomicMax(T* p, T v) {
  T old = *p;
  if (v > old) *p = v;
  return old;
}

#endif
