/*
 * wga_rt.h — the few runtime calls the C-ABI needs, in two builds:
 *   default : HIP (hipcc --offload-arch=gfx950); this is the product.
 *   WGA_EMU : g++ + tests/emu/simt_emu.h; kernel-logic emulation for the CPU test-suite only.
 */
#ifndef WGA_RT_H
#define WGA_RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef WGA_EMU
#include <stdlib.h>
#include <string.h>

#include "simt_emu.h"
/* A stream of the emulator runs everything at once, in call order; it only keeps the book that lets a test ask what WOULD be in
 * flight together on hardware: a peer copy stays "outstanding" on its stream until that stream gets its next piece of work,
 * is synchronised, or an event recorded behind the copy is waited for (rt_peer_stats). */
struct emu_stream {
  int pending_dst = -1; /* device an outstanding peer copy writes to */
};
typedef emu_stream* wga_stream_t;
struct emu_peer_book {
  int outstanding[64] = {0}, most[64] = {0};
};
static inline emu_peer_book& emu_book() {
  static emu_peer_book b;
  return b;
}
static inline void emu_stream_retire(wga_stream_t s) {
  if (s && s->pending_dst >= 0) {
    emu_book().outstanding[s->pending_dst & 63]--;
    s->pending_dst = -1;
  }
}
#define WGA_LAUNCH(kernel, grid, block, stream, ...)                         \
  do {                                                                       \
    emu_stream_retire(stream);                                               \
    emu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); });    \
  } while (0)
/* WGA_EMU_DEVICES=N: the emulator reports N devices (contexts are independent host-memory arenas), so that the CPU
 * test-suite can drive the host layer's multi-device paths (`wgatools --gpus N`) */
static inline int rt_device_count() {
  const char* e = getenv("WGA_EMU_DEVICES");
  const int n = e ? atoi(e) : 1;
  return n >= 1 && n <= 64 ? n : 1;
}
static inline const char* rt_set_device(int) { return nullptr; }
static inline const char* rt_stream_create(wga_stream_t* s) {
  *s = new emu_stream();
  return nullptr;
}
static inline void rt_stream_destroy(wga_stream_t s) {
  emu_stream_retire(s);
  delete s;
}
static inline const char* rt_sync(wga_stream_t s) {
  emu_stream_retire(s);
  return nullptr;
}
static inline const char* rt_malloc(void** p, size_t n) {
  *p = malloc(n + 16); /* a kernel may load the aligned 16-byte group that holds an array's last element */
  return *p ? nullptr : "malloc failed";
}
static inline const char* rt_free(void* p) {
  free(p);
  return nullptr;
}
static inline const char* rt_h2d(void* d, const void* h, size_t n, wga_stream_t s) {
  emu_stream_retire(s);
  memcpy(d, h, n);
  return nullptr;
}
static inline const char* rt_d2h(void* h, const void* d, size_t n, wga_stream_t s) {
  emu_stream_retire(s);
  memcpy(h, d, n);
  return nullptr;
}
static inline const char* rt_memset(void* d, int v, size_t n, wga_stream_t s) {
  emu_stream_retire(s);
  memset(d, v, n);
  return nullptr;
}
static inline const char* rt_host_alloc(void** p, size_t n) {
  *p = malloc(n ? n : 1);
  return *p ? nullptr : "malloc failed";
}
static inline const char* rt_host_free(void* p) {
  free(p);
  return nullptr;
}
static inline const char* rt_d2h_async(void* h, const void* d, size_t n, wga_stream_t s) {
  emu_stream_retire(s);
  memcpy(h, d, n);
  return nullptr;
}
static inline const char* rt_peer_copy(void* dst, int dst_dev, const void* src, int, size_t n, wga_stream_t s) {
  emu_stream_retire(s); /* a stream runs its copies one after the other */
  memcpy(dst, src, n);
  if (s) {
    emu_peer_book& b = emu_book();
    s->pending_dst = dst_dev & 63;
    if (++b.outstanding[dst_dev & 63] > b.most[dst_dev & 63]) b.most[dst_dev & 63] = b.outstanding[dst_dev & 63];
  }
  return nullptr;
}
static inline const char* rt_peer_enable(int, int) { return nullptr; } /* every "device" of the emulator is host memory */
static inline const char* rt_launch_error() { return nullptr; }
struct emu_event {
  wga_stream_t on = nullptr;
};
typedef emu_event* rt_event_t;
static inline const char* rt_event_create(rt_event_t* e) {
  *e = new emu_event();
  return nullptr;
}
static inline void rt_event_destroy(rt_event_t e) { delete e; }
static inline const char* rt_event_record(rt_event_t e, wga_stream_t s) {
  e->on = s;
  return nullptr;
}
static inline const char* rt_stream_wait_event(wga_stream_t, rt_event_t e) {
  emu_stream_retire(e->on); /* what was enqueued in front of the event has happened */
  return nullptr;
}
static inline const char* rt_event_elapsed_ms(rt_event_t, rt_event_t, float* ms) {
  *ms = 0.0f;
  return nullptr;
}
#else
#include <hip/hip_runtime.h>
typedef hipStream_t wga_stream_t;
#define WGA_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
static inline const char* rt_err(hipError_t e) { return e == hipSuccess ? nullptr : hipGetErrorString(e); }
static inline int rt_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
static inline const char* rt_set_device(int d) { return rt_err(hipSetDevice(d)); }
static inline const char* rt_stream_create(wga_stream_t* s) {
  return rt_err(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
}
static inline void rt_stream_destroy(wga_stream_t s) { (void)hipStreamDestroy(s); }
static inline const char* rt_sync(wga_stream_t s) { return rt_err(hipStreamSynchronize(s)); }
static inline const char* rt_malloc(void** p, size_t n) { return rt_err(hipMalloc(p, n ? n : 1)); }
static inline const char* rt_free(void* p) { return rt_err(hipFree(p)); }
static inline const char* rt_h2d(void* d, const void* h, size_t n, wga_stream_t s) {
  return rt_err(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
}
static inline const char* rt_d2h(void* h, const void* d, size_t n, wga_stream_t s) {
  hipError_t e = hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return hipGetErrorString(e);
  return rt_err(hipStreamSynchronize(s));
}
static inline const char* rt_memset(void* d, int v, size_t n, wga_stream_t s) {
  return rt_err(hipMemsetAsync(d, v, n, s));
}
static inline const char* rt_host_alloc(void** p, size_t n) { return rt_err(hipHostMalloc(p, n ? n : 1, hipHostMallocDefault)); }
static inline const char* rt_host_free(void* p) { return rt_err(hipHostFree(p)); }
static inline const char* rt_d2h_async(void* h, const void* d, size_t n, wga_stream_t s) {
  return rt_err(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
}
static inline const char* rt_peer_copy(void* dst, int dst_dev, const void* src, int src_dev, size_t n, wga_stream_t s) {
  return rt_err(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, n, s));
}
/* device `dev` may read `peer`'s memory from its kernels (xGMI); nullptr when it can (already enabled counts), else why not */
static inline const char* rt_peer_enable(int dev, int peer) {
  if (dev == peer) return nullptr;
  int can = 0;
  hipError_t e = hipDeviceCanAccessPeer(&can, dev, peer);
  if (e != hipSuccess) return hipGetErrorString(e);
  if (!can) return "no peer access between the two devices";
  if ((e = hipSetDevice(dev)) != hipSuccess) return hipGetErrorString(e);
  e = hipDeviceEnablePeerAccess(peer, 0);
  if (e == hipErrorPeerAccessAlreadyEnabled) {
    (void)hipGetLastError();
    return nullptr;
  }
  return rt_err(e);
}
static inline const char* rt_launch_error() { return rt_err(hipGetLastError()); }
typedef hipEvent_t rt_event_t;
static inline const char* rt_event_create(rt_event_t* e) { return rt_err(hipEventCreate(e)); }
static inline void rt_event_destroy(rt_event_t e) { (void)hipEventDestroy(e); }
static inline const char* rt_event_record(rt_event_t e, wga_stream_t s) { return rt_err(hipEventRecord(e, s)); }
static inline const char* rt_stream_wait_event(wga_stream_t s, rt_event_t e) { return rt_err(hipStreamWaitEvent(s, e, 0)); }
static inline const char* rt_event_elapsed_ms(rt_event_t a, rt_event_t b, float* ms) {
  hipError_t r = hipEventSynchronize(b);
  if (r != hipSuccess) return hipGetErrorString(r);
  return rt_err(hipEventElapsedTime(ms, a, b));
}
#endif

#endif
