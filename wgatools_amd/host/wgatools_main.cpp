/*
 * wgatools_main.cpp — the `wgatools <subcmd>` command line for the subcommands on the CIGAR hot
 * path, driving libwgahip.so through the C-ABI (include/wga_hip.h).  Flags, aliases, output text
 * and error behaviour follow the reference CLI (cli.rs:20-36,39-325; main.rs:14-206;
 * utils.rs wrap_*).  Subcommands off the hot path are not provided.
 *
 *   paf2maf | p2m   converter.rs:176-265      stat | st      tools/stat.rs:61-126
 *   maf2paf | m2p   converter.rs:29-54        pafcov | pc    tools/pafcov.rs:13-83
 */
#include <errno.h>
#include <fcntl.h>

#include <atomic>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <regex>
#include <thread>

#include "wga_host.hpp"

using namespace wga;

namespace {

/* ---- device helper ---------------------------------------------------------------------------- */
/* WGA_TIMING=1: wall time per phase of a command on stderr when it ends (profiles/r02_cli_e2e.txt) */
static double g_warm_seconds = 0.0;   /* what the HIP start-up thread took by itself */
static double g_reader_seconds = 0.0; /* what the read-ahead thread spent inside the reader */
struct PhaseTimer {
  bool on = getenv("WGA_TIMING") != nullptr;
  std::vector<std::pair<std::string, double>> acc;
  double last = now();
  double t_start = last;
  bool printed = false;
  static double now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
  }
  std::thread::id owner = std::this_thread::get_id();
  void mark(const char* name) { /* the time since the previous mark belongs to `name` (main thread only: device workers do not mark) */
    if (!on || std::this_thread::get_id() != owner) return;
    const double t = now();
    for (auto& a : acc)
      if (a.first == name) {
        a.second += t - last;
        last = t;
        return;
      }
    acc.emplace_back(name, t - last);
    last = t;
  }
  ~PhaseTimer() { print(); }
  void print() {
    if (!on || printed) return;
    printed = true;
    double tot = 0;
    for (auto& a : acc) tot += a.second;
    fprintf(stderr, "[timing]");
    for (auto& a : acc) fprintf(stderr, " %s %.3f s |", a.first.c_str(), a.second);
    fprintf(stderr, " marked total %.3f s", tot);
    if (g_warm_seconds > 0.0) fprintf(stderr, " | hip start-up thread %.3f s", g_warm_seconds);
    if (g_reader_seconds > 0.0) fprintf(stderr, " | reader thread %.3f s", g_reader_seconds);
    fprintf(stderr, " | since the process's static initialisers %.3f s\n", now() - t_start);
  }
};
static PhaseTimer g_timer;

/* A command whose outputs are closed leaves the process at once. Returning through the destructors unmapped a multi-gigabyte
 * input, handed gigabytes of pinned memory back and took the HIP runtime down page by page: 0.3-0.5 s of a 1.4-2 s run
 * (profiles/r05_e2e_at_size.json: `stat -f maf` 0.88 s of marked phases in a 1.36 s process) for memory the kernel reclaims
 * anyway. Every output of every command goes through Output::close() (fclose / gzclose) before this is reached; the streams
 * the process was given are flushed here. The error paths still unwind normally. */
static int leave(int rc) {
  g_timer.mark("teardown before leaving");
  g_timer.print();
  fflush(stdout);
  fflush(stderr);
  _exit(rc);
}

/* The HIP runtime takes ~0.1 s to come up: main() starts it on a second thread while the command opens and reads its
 * input; the first Dev::init() picks the context up (one context per process: the commands use one Dev). */
struct GpuWarm {
  std::thread th;
  wga_ctx* ctx = nullptr;
  int rc = 0;
  std::string err;
  bool started = false, taken = false;
  void start() {
    started = true;
    th = std::thread([this] {
      const double t0 = PhaseTimer::now();
      rc = wga_ctx_create(0, &ctx);
      if (rc) {
        err = wga_last_error();
        return;
      }
      void* warm = nullptr; /* the first allocation pays for the runtime's lazy initialisation */
      if (wga_malloc(ctx, 256, &warm) == 0) wga_free(ctx, warm);
      g_warm_seconds = PhaseTimer::now() - t0;
    });
  }
  ~GpuWarm() {
    if (th.joinable()) th.join();
    if (started && !taken && ctx) wga_ctx_destroy(ctx);
  }
};
static GpuWarm g_warm;

/* Streams n bytes of a device buffer into the output: pinned staging buffers, the copy of piece k + 1 runs while
 * piece k is written; into a plain file the pieces are written with pwrite by a few threads (one write() stream into
 * the page cache moves 2-4 GB/s, the copy engine > 40 GB/s), anything else (stdout, .gz) keeps the one ordered
 * writer.  Replaces "download everything into one std::string, then fwrite" (r01: paf2maf 1.8 s for 3 GB of MAF). */
struct DevStreamer {
  static const size_t kPiece = (size_t)16 << 20;
  static const int kBufs = 40; /* at most; a run takes what its writers need (below) */
  wga_ctx* ctx;
  void* buf[kBufs];
  explicit DevStreamer(wga_ctx* c) : ctx(c) {
    for (int k = 0; k < kBufs; k++) buf[k] = nullptr;
  }
  ~DevStreamer() {
    for (int k = 0; k < kBufs; k++)
      if (buf[k]) wga_host_free(ctx, buf[k]);
  }
  /* raw: d_src holds finished BGZF members for a `.gz` file (Output::raw_fd) */
  void run(Output& out, const uint8_t* d_src, size_t n, bool raw = false) {
    if (n == 0) return;
    const size_t np = (n + kPiece - 1) / kPiece;
    /* writers: memcpy into the page cache is what bounds a large output (one thread moves 2-3 GB/s), the copy engine delivers
     * far more; WGA_WRITE_THREADS overrides */
    int nthreads = 1;
    {
      uint64_t pos_probe = 0;
      if ((raw ? out.raw_fd(&pos_probe) : out.plain_fd(&pos_probe)) >= 0) {
        nthreads = 8; /* 16 / 24 / 32 measured slower at 15 GB (scripts/gpu_write_threads.py: 2.6 / 2.9 / 3.0 s against 2.5) */
        if (const char* e = getenv("WGA_WRITE_THREADS")) nthreads = std::max(1, std::min(32, atoi(e)));
      }
    }
    const int nbuf = (int)std::min<size_t>((size_t)std::min(kBufs, nthreads + 4), np + 1); /* pinned memory is slow to get: only what this run uses */
    for (int k = 0; k < nbuf; k++)
      if (!buf[k] && wga_host_alloc(ctx, kPiece, &buf[k])) fail(std::string("GPU engine: ") + wga_last_error());
    uint64_t pos0 = 0;
    const int fd = raw ? out.raw_fd(&pos0) : out.plain_fd(&pos0);
    if (raw && fd < 0) fail("internal error: BGZF members for something that is not a .gz file");
    /* a plain file: its blocks are allocated once, up front (the system call, not posix_fallocate: no emulation by writing where
     * a file system lacks it).  Eight threads writing the same 16 MB into a NEW file: 12.6 GB/s, 16.3 after fallocate
     * (profiles/r04_cli_e2e.txt); with real pieces out of the pinned buffers paf2maf's 15 GB leave at 10-12 GB/s either way,
     * and neither 12-24 writers nor handing the pieces out as eight sequential streams changed that beyond the run-to-run spread. */
    /* FALLOC_FL_KEEP_SIZE: the blocks are reserved, the file's length still ends behind the last byte written — a run that fails
     * or is killed half way leaves no zero-filled tail that looks like output. */
    if (fd >= 0 && n >= ((size_t)64 << 20)) (void)fallocate(fd, FALLOC_FL_KEEP_SIZE, (off_t)pos0, (off_t)n);
    std::vector<std::thread> writers;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> state(kBufs, 4); /* 0 free, 1 filled (piece index in `which`), 2 being written, 3 being filled, 4 not allocated */
    for (int k = 0; k < nbuf; k++) state[k] = 0;
    std::vector<size_t> which(kBufs, 0);
    size_t next_write = 0; /* ordered writer: next piece to go out */
    bool failed = false, done_filling = false;
    std::string copy_err;
    auto writer = [&]() {
      for (;;) {
        int b = -1;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] {
            if (failed) return true;
            for (int k = 0; k < kBufs; k++)
              if (state[k] == 1 && (fd >= 0 || which[k] == next_write)) return true;
            return done_filling;
          });
          if (failed) return;
          for (int k = 0; k < kBufs; k++)
            if (state[k] == 1 && (fd >= 0 || which[k] == next_write)) {
              b = k;
              state[k] = 2;
              break;
            }
          if (b < 0) return; /* done_filling and nothing left for this thread */
        }
        const size_t p = which[b], off = p * kPiece, len = std::min(kPiece, n - off);
        bool ok = true;
        if (fd >= 0) {
          size_t w = 0;
          while (w < len) {
            const ssize_t r = pwrite(fd, (const char*)buf[b] + w, len - w, (off_t)(pos0 + off + w));
            if (r <= 0) {
              ok = false;
              break;
            }
            w += (size_t)r;
          }
        } else {
          try {
            out.write((const char*)buf[b], len);
          } catch (...) {
            ok = false;
          }
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          state[b] = 0;
          if (fd < 0) next_write++;
          if (!ok) failed = true;
        }
        cv.notify_all();
      }
    };
    for (int t = 0; t < nthreads; t++) writers.emplace_back(writer);
    /* the filler: issue the copy of a piece into a free buffer, wait for it, hand it over */
    for (size_t p = 0; p < np && !failed; p++) {
      int b = -1;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] {
          if (failed) return true;
          for (int k = 0; k < kBufs; k++)
            if (state[k] == 0) return true;
          return false;
        });
        if (failed) break;
        for (int k = 0; k < kBufs; k++)
          if (state[k] == 0) {
            b = k;
            state[k] = 3; /* being filled */
            break;
          }
      }
      const size_t off = p * kPiece, len = std::min(kPiece, n - off);
      if (wga_memcpy_d2h_async(ctx, buf[b], d_src + off, len) || wga_sync(ctx)) {
        /* the writers are joinable threads: stop and join them before the error leaves this frame (unwinding past a
         * joinable std::thread is std::terminate, and the message would never be printed) */
        copy_err = std::string("GPU engine: ") + wga_last_error();
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        break;
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        which[b] = p;
        state[b] = 1;
      }
      cv.notify_all();
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      done_filling = true;
    }
    cv.notify_all();
    for (auto& t : writers) t.join();
    if (!copy_err.empty()) fail(copy_err);
    if (failed) fail("IO error:write failed");
    if (fd >= 0) out.advance(n);
  }
};

/* `wgatools --gpus N` (ours, outside the reference's flag namespace: SURVEY.md section 5): the PAF commands paf2maf, stat and
 * pafcov shard their records over N devices by fnv1a64(target_name) % N — one worker thread and one context per device */
static int g_gpus = 1;
static bool g_spread = false; /* `--spread` (with --gpus N, pafcov): deal the records out round robin and sum the coverage over the devices */

struct Dev;
static Dev* g_gz_dev = nullptr; /* the device host text for a `.gz` file is deflated on (Output::big_text) */
static bool big_text_through_device(Output& out, const char* p, size_t n);

struct Dev {
  wga_ctx* ctx = nullptr;
  bool own_ctx = true; /* false: the context is another Dev's (the reader's, lent to device 0's worker) */
  int device = 0;
  bool helper = false; /* a reader's own context, used on its helper thread (BgzfDeviceSource): never the one host text is deflated on */
  std::vector<void*> owned;
  std::unique_ptr<DevStreamer> streamer; /* pinned buffers, allocated once per process */
  Dev() {}
  explicit Dev(int dev) : device(dev) {}
  void init() {
    /* the device big host text for a `.gz` file goes through (Output::big_text): a COMMAND's context on device 0, registered by the
     * thread that runs the command — never a helper thread's (a wga_ctx serves one thread), and again whenever the slot is empty
     * (a reader that came and went leaves it so) */
    if (!helper && device == 0 && !g_gz_dev) {
      g_gz_dev = this;
      Output::big_text = big_text_through_device;
    }
    if (ctx) return;
    g_timer.mark("host");
    if (device != 0) { /* the workers' devices: a context each */
      int rc = wga_ctx_create(device, &ctx);
      if (rc) fail(std::string("GPU engine: ") + wga_last_error());
      return;
    }
    if (g_warm.started && !g_warm.taken) {
      if (g_warm.th.joinable()) g_warm.th.join();
      g_warm.taken = true;
      if (g_warm.rc) fail("GPU engine: " + g_warm.err);
      ctx = g_warm.ctx;
      g_timer.mark("hip init (what was left of it)");
      return;
    }
    int rc = wga_ctx_create(0, &ctx);
    if (rc) fail(std::string("GPU engine: ") + wga_last_error());
    void* warm = nullptr;
    if (wga_malloc(ctx, 256, &warm) == 0) wga_free(ctx, warm);
    g_timer.mark("hip init");
  }
  void check(int rc) {
    if (rc) fail(std::string("GPU engine: ") + wga_last_error());
  }
  void* alloc(size_t bytes) {
    void* p = nullptr;
    check(wga_malloc(ctx, bytes ? bytes : 16, &p));
    owned.push_back(p);
    return p;
  }
  template <typename T>
  T* upload(const T* h, size_t n) {
    T* d = (T*)alloc(n * sizeof(T));
    if (n) check(wga_memcpy_h2d(ctx, d, h, n * sizeof(T)));
    return d;
  }
  template <typename T>
  T* upload(const std::vector<T>& v) { return upload(v.data(), v.size()); }
  template <typename T>
  void download(T* h, const T* d, size_t n) { check(wga_memcpy_d2h(ctx, h, d, n * sizeof(T))); }
  /* The rows of paf2maf go into ONE device buffer for the whole run (it grows when a piece needs more): a buffer per piece
   * would pay the allocation and the first touch of gigabytes every time. */
  void* out_arena = nullptr;
  size_t out_arena_cap = 0;
  void out_arena_for(size_t bytes) {
    if (bytes <= out_arena_cap) return;
    if (out_arena) {
      check(wga_sync(ctx));
      wga_free(ctx, out_arena);
      out_arena = nullptr;
      out_arena_cap = 0;
    }
    const size_t cap = bytes + bytes / 4;
    check(wga_malloc(ctx, cap, &out_arena));
    out_arena_cap = cap;
  }
  /* the BGZF members of a piece of `.gz` output (K18) wait here for the copy out */
  void* gz_arena = nullptr;
  size_t gz_arena_cap = 0;
  void gz_arena_for(size_t bytes) {
    if (bytes <= gz_arena_cap) return;
    if (gz_arena) {
      check(wga_sync(ctx));
      wga_free(ctx, gz_arena);
      gz_arena = nullptr;
      gz_arena_cap = 0;
    }
    check(wga_malloc(ctx, bytes, &gz_arena));
    gz_arena_cap = bytes;
  }
  void release(void* p) {
    auto it = std::find(owned.begin(), owned.end(), p);
    if (it != owned.end()) owned.erase(it);
    wga_free(ctx, p);
  }
  void release_all() {
    check(wga_sync(ctx));
    for (void* p : owned) wga_free(ctx, p);
    owned.clear();
  }
  void release_to(size_t keep) { /* frees everything allocated after the first `keep` buffers */
    check(wga_sync(ctx));
    while (owned.size() > keep) {
      wga_free(ctx, owned.back());
      owned.pop_back();
    }
  }
  ~Dev() {
    if (g_gz_dev == this) g_gz_dev = nullptr;
    if (ctx) {
      streamer.reset();
      if (out_arena) wga_free(ctx, out_arena);
      if (gz_arena) wga_free(ctx, gz_arena);
      for (void* p : owned) wga_free(ctx, p);
      if (own_ctx) wga_ctx_destroy(ctx);
    }
  }
};
/* marks: the phase timer is the main thread's; a helper thread that streams a piece out while the next one is worked on passes false */
static void stream_out(Dev& d, Output& out, const uint8_t* d_src, size_t n, bool marks = true) {
  if (!d.streamer) d.streamer.reset(new DevStreamer(d.ctx));
  d.check(wga_sync(d.ctx));
  if (marks) g_timer.mark("kernels + host tables");
  if (out.bgzf) {
    /* `.gz`: the text is deflated where it is (K18) and the members are what crosses PCIe and reaches the file; slabs keep
     * the buffer of members bounded whatever the piece */
    const size_t kSlab = (size_t)1 << 30;
    for (size_t a = 0; a < n; a += kSlab) {
      const size_t len = std::min(kSlab, n - a);
      const uint64_t cap = wga_bgzf_bound(len);
      d.gz_arena_for((size_t)cap);
      uint64_t used = 0;
      d.check(wga_bgzf_compress(d.ctx, d_src + a, len, (uint8_t*)d.gz_arena, cap, &used, 0));
      if (marks) g_timer.mark("device deflate");
      d.streamer->run(out, (const uint8_t*)d.gz_arena, (size_t)used, true);
      if (marks) g_timer.mark("copy out + write");
    }
    return;
  }
  d.streamer->run(out, d_src, n);
  if (marks) g_timer.mark("copy out + write");
}
/* host text of megabytes for a `.gz` file (Output::big_text): up, through the same deflate, out */
static bool big_text_through_device(Output& out, const char* p, size_t n) {
  Dev* d = g_gz_dev;
  if (!d || !d->ctx) return false;
  const size_t kSlab = (size_t)1 << 30;
  for (size_t a = 0; a < n; a += kSlab) {
    const size_t len = std::min(kSlab, n - a);
    void* d_txt = nullptr;
    d->check(wga_malloc(d->ctx, len + 16, &d_txt));
    d->check(wga_memcpy_h2d(d->ctx, d_txt, p + a, len));
    stream_out(*d, out, (const uint8_t*)d_txt, len);
    d->check(wga_sync(d->ctx));
    wga_free(d->ctx, d_txt);
  }
  return true;
}

/* A bgzipped PAF / MAF through the device inflate (K17, wga_bgzf_inflate): the compressed file stays on the host, a run of
 * members at a time (64 MiB of text) is uploaded, inflated one wave per member and read back; LineChunkReader takes the bytes
 * where gzread would have produced them.  The producer has a context of its own on device 0 (it runs on the reader's helper
 * thread).  WGA_BGZF_DEVICE=0, a plain gzip stream or stdin keep zlib. */
struct BgzfDeviceSource {
  std::string img, path; /* img: the compressed bytes of the run of members being inflated (never the file) */
  std::vector<BgzfMember> members;
  uint64_t total = 0, file_bytes = 0;
  size_t next_member = 0;
  std::string buf;
  size_t buf_at = 0;
  int fd = -1;
  uint64_t kBatch = 64ull << 20; /* text bytes per run of members (WGA_BGZF_BATCH: the tests reach several runs with small files) */
  Dev d;
  ~BgzfDeviceSource() {
    if (fd >= 0) ::close(fd);
  }
  bool open(const std::string* p) {
    d.helper = true; /* refill() runs on the reader's helper thread: this context is no one else's */
    const char* e = getenv("WGA_BGZF_DEVICE");
    if (!p || (e && atoi(e) == 0)) return false;
    if (!scan_bgzf_members(*p, members, &total, &file_bytes)) return false;
    path = *p;
    fd = ::open(p->c_str(), O_RDONLY);
    if (fd < 0) fail("File path `" + *p + "` not exist");
    if (const char* b = getenv("WGA_BGZF_BATCH")) kBatch = std::max<uint64_t>(1, strtoull(b, nullptr, 10));
    return true;
  }
  bool refill() {
    if (next_member == members.size()) return false;
    const size_t m0 = next_member;
    size_t m1 = m0;
    uint64_t out_bytes = 0;
    while (m1 < members.size() && (m1 == m0 || out_bytes + members[m1].out_len <= kBatch)) out_bytes += members[m1++].out_len;
    const uint64_t c0 = members[m0].in_off, c1 = members[m1 - 1].in_off + members[m1 - 1].in_len, o0 = members[m0].out_off;
    std::vector<BgzfMember> part(members.begin() + m0, members.begin() + m1);
    for (BgzfMember& k : part) {
      k.in_off -= c0;
      k.out_off -= o0;
    }
    static_assert(sizeof(BgzfMember) == sizeof(wga_bgzf_block), "wga_bgzf_block layout");
    { /* this run's compressed bytes, the last member's trailer and the slack the kernel's whole-vector loads reach into */
      const size_t want = (size_t)std::min<uint64_t>(c1 - c0 + 8, file_bytes - c0);
      img.assign((size_t)(c1 - c0) + 16, '\0');
      size_t done = 0;
      while (done < want) {
        const ssize_t r = pread(fd, &img[done], want - done, (off_t)(c0 + done));
        if (r <= 0) fail("IO error:short read of `" + path + "`");
        done += (size_t)r;
      }
    }
    d.init();
    const uint8_t* d_img = d.upload((const uint8_t*)img.data(), (size_t)(c1 - c0) + 16);
    const wga_bgzf_block* d_mem = (const wga_bgzf_block*)d.upload(part.data(), part.size());
    uint8_t* d_text = (uint8_t*)d.alloc(out_bytes + 64);
    auto* d_st = (uint32_t*)d.alloc(part.size() * 4 + 4);
    d.check(wga_bgzf_inflate(d.ctx, d_img, (uint64_t)(c1 - c0), (uint32_t)part.size(), d_mem, d_text, d_st));
    std::vector<uint32_t> st(part.size());
    d.download(st.data(), (const uint32_t*)d_st, st.size());
    for (uint32_t v : st)
      if (v) fail("IO error:corrupt BGZF block in `" + path + "`");
    buf.resize((size_t)out_bytes);
    if (out_bytes) d.download((uint8_t*)&buf[0], d_text, out_bytes);
    d.release_all();
    { /* the members' CRC-32 (the trailer behind the deflate data) as gzread checks it: host threads over the members */
      const unsigned T = 8;
      std::vector<std::thread> th;
      std::vector<int> bad(T, 0);
      for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t] {
          for (size_t k = t; k < part.size(); k += T) {
            const unsigned char* tr = (const unsigned char*)img.data() + part[k].in_off + part[k].in_len;
            const uint32_t want = (uint32_t)tr[0] | (uint32_t)tr[1] << 8 | (uint32_t)tr[2] << 16 | (uint32_t)tr[3] << 24;
            if (gzip_crc32(buf.data() + part[k].out_off, part[k].out_len) != want) bad[t] = 1;
          }
        });
      for (auto& x : th) x.join();
      for (int v : bad)
        if (v) fail("IO error:" + path + ": incorrect data check");
    }
    buf_at = 0;
    next_member = m1;
    return true;
  }
  size_t read(char* dst, size_t want) {
    while (buf_at == buf.size())
      if (!refill()) return 0;
    const size_t n = std::min(want, buf.size() - buf_at);
    memcpy(dst, buf.data() + buf_at, n);
    buf_at += n;
    return n;
  }
};

/* An indexed FASTA whose sequence pool lives in HBM.  The file — plain, gzip or BGZF (inflated on
 * all host cores) — is uploaded as text, wga_fasta_pool strips the line ends on the device and returns the contig table;
 * the drivers then address (contig, start, length) as pool offsets with htslib's clipping (Faidx::fetch) and never copy
 * a slice.  Commands that need bases on the host (VCF REF / ALT text, a target row of pafpseudo, the offending base of
 * an error message) read them back from the device.  WGA_FASTA_READER=host keeps the host line stripper (A/B, tests). */
struct DevFasta {
  /* The pool the kernels see has kPad bytes of 'N' in front of the first contig and behind the last one: the streaming row
   * kernel reads 16-byte windows that may reach over a slice's ends and leaves slices within 32 bytes of a pool edge to the
   * block kernels (include/wga_hip.h) — with the padding no slice of a real contig is such a slice.  Offsets handed out by
   * fetch() count from the padded pool's start. */
  static constexpr uint64_t kPad = 32;
  Faidx idx;
  uint8_t* d_pool = nullptr;
  uint64_t bytes = 0;
  bool have_host = false;
  void load(Dev& d, const std::string& path) {
    const char* mode = getenv("WGA_FASTA_READER");
    if (mode && strcmp(mode, "host") == 0) {
      idx.load(path);
      idx.pool.insert(0, (size_t)kPad, 'N');
      idx.pool.append((size_t)kPad, 'N');
      bytes = idx.pool.size();
      d_pool = d.upload((const uint8_t*)idx.pool.data(), idx.pool.size());
      have_host = true;
      return;
    }
    /* a bgzipped FASTA goes to the device compressed and is inflated there, block by block (wga_bgzf_inflate): the text never
     * exists on the host; WGA_BGZF_DEVICE=0 keeps the host threads' inflate */
    std::string text;
    uint8_t* d_text = nullptr;
    uint64_t n_text = 0;
    bool on_device_text = false;
    {
      const char* e = getenv("WGA_BGZF_DEVICE");
      std::string img;
      std::vector<BgzfMember> members;
      if (!(e && atoi(e) == 0) && read_bgzf_image(path, img, members, &n_text) && n_text < 0xFFFFFFF0ull) {
        static_assert(sizeof(BgzfMember) == sizeof(wga_bgzf_block), "wga_bgzf_block layout");
        img.append(16, '\0');
        const uint8_t* d_img = d.upload((const uint8_t*)img.data(), img.size());
        const wga_bgzf_block* d_mem = (const wga_bgzf_block*)d.upload(members.data(), members.size());
        d_text = (uint8_t*)d.alloc(n_text + 64);
        auto* d_st = (uint32_t*)d.alloc(members.size() * 4 + 4);
        d.check(wga_bgzf_inflate(d.ctx, d_img, img.size() - 16, (uint32_t)members.size(), d_mem, d_text, d_st));
        std::vector<uint32_t> st(members.size());
        if (!st.empty()) d.download(st.data(), (const uint32_t*)d_st, st.size());
        for (uint32_t v : st)
          if (v) fail("IO error:corrupt BGZF block in `" + path + "`");
        d.release(d_st);
        d.release((void*)d_mem);
        d.release((void*)d_img);
        on_device_text = true;
      }
    }
    if (!on_device_text) {
      text = read_all_parallel(path);
      n_text = text.size();
      d_text = d.upload((const uint8_t*)text.data(), text.size());
    }
    uint64_t nc = 0, nb = 0;
    d.check(wga_fasta_pool(d.ctx, d_text, n_text, &nc, &nb, nullptr, nullptr));
    d_pool = (uint8_t*)d.alloc(nb + 2 * kPad + 64);
    d.check(wga_memset(d.ctx, d_pool, 'N', kPad));
    d.check(wga_memset(d.ctx, d_pool + kPad + nb, 'N', kPad + 64));
    auto* d_tab = (wga_fa_contig*)d.alloc((nc + 1) * sizeof(wga_fa_contig));
    d.check(wga_fasta_pool(d.ctx, d_text, n_text, &nc, &nb, d_pool + kPad, d_tab));
    std::vector<wga_fa_contig> tab(nc);
    if (nc) d.download(tab.data(), (const wga_fa_contig*)d_tab, nc);
    static_assert(sizeof(wga_fa_contig) == 4 * sizeof(uint64_t), "wga_fa_contig layout");
    if (on_device_text) { /* the names: only the header lines come back, one behind the other */
      std::string hdrs;
      std::vector<wga_fa_contig> t2(tab);
      for (size_t k = 0; k < nc; k++) {
        const uint64_t hs = tab[k].hdr_start, he = tab[k].hdr_end;
        const size_t at = hdrs.size();
        hdrs.resize(at + (size_t)(he - hs) + 1);
        if (he > hs) d.download((uint8_t*)&hdrs[at], (const uint8_t*)d_text + hs, (size_t)(he - hs));
        hdrs[at + (size_t)(he - hs)] = '\n';
        t2[k].hdr_start = at;
        t2[k].hdr_end = at + (he - hs);
      }
      idx.set_table(hdrs, (const uint64_t*)t2.data(), nc);
    } else {
      idx.set_table(text, (const uint64_t*)tab.data(), nc);
    }
    bytes = nb + 2 * kPad;
    d.release(d_tab);
    d.release(d_text);
  }
  void fetch(const std::string& name, uint64_t beg, uint64_t end_incl, uint64_t* off, uint64_t* len) const {
    idx.fetch(name, beg, end_incl, off, len);
    *off += kPad;
  }
  /* the whole pool on the host (downloaded once) */
  const std::string& host_pool(Dev& d) {
    if (!have_host) {
      idx.pool.resize(bytes);
      if (bytes) d.download((uint8_t*)&idx.pool[0], (const uint8_t*)d_pool, bytes);
      have_host = true;
    }
    return idx.pool;
  }
  char at(Dev& d, uint64_t off) {
    if (have_host) return idx.pool[off];
    uint8_t c = 0;
    d.download(&c, (const uint8_t*)d_pool + off, 1);
    return (char)c;
  }
  std::string slice(Dev& d, uint64_t off, uint64_t len) {
    if (have_host) return idx.pool.substr(off, len);
    std::string s2(len, '\0');
    if (len) d.download((uint8_t*)&s2[0], (const uint8_t*)d_pool + off, len);
    return s2;
  }
};

/* the CIGAR text (after the tag) of record k of a batch, for error messages: owned strings, or spans
 * of the input file when the records came from the device splitter */
struct CigarTexts {
  std::vector<std::string> owned;
  const std::string* file = nullptr;
  std::vector<uint64_t> beg, end;
  void push_back(std::string s) { owned.push_back(std::move(s)); }
  const std::string& back() const { return owned.back(); }
  std::string operator[](size_t k) const { return file ? file->substr(beg[k], end[k] - beg[k]) : owned[k]; }
};

/* A PAF input: the records (paf.rs:50-65) and where their CIGARs are.  Plain files are split on the
 * device (wga_paf_split: fixed fields, name spans, the cg:Z: span; tags are not materialised) and their
 * CIGARs are tokenised where they lie in the uploaded text; a file with any line the splitter does not
 * take (quotes, CR, short or malformed lines, cs:Z: only) goes through the csv-semantics host parser. */
struct PafInput {
  std::string text;
  std::vector<PafRecord> recs;
  bool on_device = false;
  uint8_t* d_text = nullptr;
  std::vector<uint64_t> cg_beg, cg_end; /* on_device: per record, cg_beg == WGA_NONE without a cg:Z: tag */
  uint64_t cigar_bytes(size_t i) const {
    if (on_device) return cg_beg[i] == WGA_NONE ? 0 : cg_end[i] - cg_beg[i];
    uint64_t b = 0;
    for (const auto& tg : recs[i].tags) b += tg.size();
    return b;
  }
};
PafInput paf_from_text(Dev& d, std::string&& text, bool want_tags, uint64_t rec0 = 0, uint64_t line0 = 0,
                       uint64_t byte0 = 0) {
  PafInput in;
  in.text = std::move(text);
  const char* force = getenv("WGA_PAF_READER"); /* "host": always the csv-semantics reader (measurements) */
  if (!want_tags && !in.text.empty() && in.text.size() < 0xFFFFFFF0ull && !(force && strcmp(force, "host") == 0)) {
    g_timer.mark("file read");
    d.init();
    in.text.append(16, '\0'); /* slack behind the text for whole-vector loads */
    in.d_text = d.upload((const uint8_t*)in.text.data(), in.text.size());
    in.text.resize(in.text.size() - 16);
    g_timer.mark("upload");
    uint64_t n_lines = 0;
    d.check(wga_paf_split(d.ctx, in.d_text, in.text.size(), &n_lines, nullptr, 0));
    auto* d_lines = (wga_paf_line*)d.alloc((size_t)(n_lines + 1) * sizeof(wga_paf_line));
    d.check(wga_paf_split(d.ctx, in.d_text, in.text.size(), &n_lines, d_lines, n_lines));
    std::vector<wga_paf_line> lines((size_t)n_lines);
    if (n_lines) d.download(lines.data(), d_lines, (size_t)n_lines);
    d.release(d_lines);
    bool plain = true;
    size_t n_rec = 0;
    for (const wga_paf_line& L : lines) {
      if (L.status == WGA_PAF_FALLBACK) plain = false;
      if (L.status == WGA_PAF_OK) n_rec++;
    }
    if (plain) {
      in.on_device = true;
      in.recs.reserve(n_rec);
      in.cg_beg.reserve(n_rec);
      in.cg_end.reserve(n_rec);
      for (const wga_paf_line& L : lines) {
        if (L.status != WGA_PAF_OK) continue;
        PafRecord r;
        r.query_name.assign(in.text, (size_t)L.qname_off, L.qname_len);
        r.target_name.assign(in.text, (size_t)L.tname_off, L.tname_len);
        r.query_length = L.num[0];
        r.query_start = L.num[1];
        r.query_end = L.num[2];
        r.target_length = L.num[3];
        r.target_start = L.num[4];
        r.target_end = L.num[5];
        r.matches = L.num[6];
        r.block_length = L.num[7];
        r.mapq = L.num[8];
        r.neg = L.strand_neg != 0;
        in.recs.push_back(std::move(r));
        in.cg_beg.push_back(L.cg_beg);
        in.cg_end.push_back(L.cg_end);
      }
      g_timer.mark("device split + host records");
      return in;
    }
    d.release(in.d_text);
    in.d_text = nullptr;
  }
  in.recs = parse_paf(in.text, rec0, line0, byte0);
  return in;
}
PafInput load_paf(Dev& d, const std::string* input, bool want_tags) {
  return paf_from_text(d, read_all(input), want_tags);
}

/* A PAF input in pieces of about 256 MiB that end at line ends (WGA_CHUNK_BYTES overrides the size), the next piece read
 * ahead while the current one is worked on: the streaming
 * commands hold one piece of the file, and its buffers on the device, at a time. */
struct PafChunks {
  LineChunkReader rd;
  bool want_tags;
  size_t target = (size_t)192 << 20; /* configs[1]'s 1.19 GB through `stat`: 0.25-0.27 s with 192 MB or 512 MB pieces, 0.35-0.40 s with 256 MB */
  uint64_t recs_before = 0;
  /* the next piece is read by a helper thread while the caller works on the current one (reading 1 GB takes as long as
   * every kernel of the run together): `ahead` holds it with the reader's counters for that piece */
  struct Ahead {
    bool ok = false;
    std::string piece, err;
    uint64_t lines_before = 0, bytes_before = 0;
  } ahead;
  std::thread reader;
  bool started = false;
  std::unique_ptr<BgzfDeviceSource> bgzf; /* a bgzipped input: inflated on the device */
  PafChunks(const std::string* input, bool tags) : want_tags(tags) {
    bgzf.reset(new BgzfDeviceSource());
    if (bgzf->open(input))
      rd.source = [this](char* dst, size_t want) { return bgzf->read(dst, want); };
    else
      bgzf.reset();
    rd.open(input);
    if (const char* e = getenv("WGA_CHUNK_BYTES")) target = (size_t)strtoull(e, nullptr, 10);
    if (target == 0) target = 1;
  }
  ~PafChunks() {
    if (reader.joinable()) reader.join();
  }
  void read_ahead() {
    reader = std::thread([this] {
      Ahead a;
      const double t0 = PhaseTimer::now();
      try {
        a.ok = rd.next(a.piece, target);
        g_reader_seconds += PhaseTimer::now() - t0; /* one reader thread at a time */
        a.lines_before = rd.lines_before;
        a.bytes_before = rd.bytes_before;
      } catch (Error& e) {
        a.err = e.msg.empty() ? std::string("error") : e.msg;
      } catch (std::exception& e) {
        a.err = std::string("internal error: ") + e.what();
      }
      ahead = std::move(a);
    });
  }
  /* the next piece with at least one record, or false at the end of the input */
  bool next(Dev& d, PafInput& in) {
    for (;;) {
      if (!started) {
        started = true;
        read_ahead();
      }
      if (!reader.joinable()) return false; /* the end was seen */
      reader.join();
      Ahead a = std::move(ahead);
      if (!a.err.empty()) fail(a.err);
      if (!a.ok) return false;
      rd.recycle(std::move(in.text)); /* the piece the caller has finished with: its buffer takes a later piece */
      in.text = std::string();
      read_ahead();
      in = paf_from_text(d, std::move(a.piece), want_tags, recs_before, a.lines_before, a.bytes_before);
      recs_before += in.recs.size();
      if (!in.recs.empty()) return true;
      if (in.d_text) d.release(in.d_text);
    }
  }
};

/* CIGAR texts of a run of records -> device batch through the device tokeniser (wga_cigar_tokenise):
 * the host only finds the tag; digits and op chars are parsed on the GPU.  Returns the reference's
 * message for the first failing record in input order ("" if none). */
std::string device_tokenise(Dev& d, const PafInput& in, size_t first, uint32_t n, CigarTexts& cigars,
                            wga_cigar_batch* cb, std::vector<wga_tok_err>* all_errs = nullptr,
                            const size_t* which = nullptr, const uint8_t* d_text_here = nullptr) {
  /* which: the records are in.recs[which[first + k]] (a device worker's share of the piece) instead of in.recs[first + k];
   * d_text_here: the piece's text on THIS device (the splitter ran on device 0) */
  auto rec_at = [&](uint32_t k) -> size_t { return which ? which[first + k] : first + k; };
  std::string blob, first_err;
  std::vector<uint64_t> toff{0};
  std::vector<uint8_t> strand;
  uint32_t n_ok = n;
  for (uint32_t k = 0; k < n; k++) {
    if (in.on_device) {
      if (in.cg_beg[rec_at(k)] == WGA_NONE) { /* errors.rs:57: only the records before it can fail earlier */
        first_err = "CIGAR start tag not found";
        n_ok = k;
        break;
      }
      cigars.beg.push_back(in.cg_beg[rec_at(k)]);
      cigars.end.push_back(in.cg_end[rec_at(k)]);
    } else {
      int err = 0;
      std::string cg = paf_cigar_string(in.recs[rec_at(k)], &err);
      if (err) {
        first_err = "CIGAR start tag not found";
        n_ok = k;
        break;
      }
      cigars.push_back(cg.substr(5));
      blob += cigars.back();
      toff.push_back(blob.size());
    }
    strand.push_back(in.recs[rec_at(k)].neg ? 1 : 0);
  }
  if (in.on_device) cigars.file = &in.text;
  cb->n = n_ok;
  cb->n_ops = 0;
  cb->d_ops = nullptr;
  cb->d_op_off = nullptr;
  cb->d_strand_neg = nullptr;
  if (n_ok == 0) return first_err;
  const uint8_t* d_text;
  const uint64_t *d_beg, *d_end;
  if (in.on_device) {
    d_text = d_text_here ? d_text_here : in.d_text;
    d_beg = d.upload(cigars.beg);
    d_end = d.upload(cigars.end);
  } else {
    blob.append(64, '0'); /* slack behind the last text */
    d_text = d.upload((const uint8_t*)blob.data(), blob.size());
    d_beg = d.upload(toff);
    d_end = d_beg + 1;
  }
  auto* d_cnt = (uint64_t*)d.alloc((size_t)n_ok * 8);
  auto* d_err = (wga_tok_err*)d.alloc((size_t)n_ok * sizeof(wga_tok_err));
  d.check(wga_cigar_tokenise_spans(d.ctx, n_ok, d_text, d_beg, d_end, d_cnt, d_err, nullptr, nullptr));
  auto* d_ooff = (uint64_t*)d.alloc(((size_t)n_ok + 1) * 8);
  d.check(wga_exclusive_scan_u64(d.ctx, n_ok, d_cnt, d_ooff));
  uint64_t total = 0;
  d.download(&total, d_ooff + n_ok, 1);
  auto* d_ops = (uint32_t*)d.alloc((total + 4) * 4);
  d.check(wga_cigar_tokenise_spans(d.ctx, n_ok, d_text, d_beg, d_end, d_cnt, d_err, d_ops, d_ooff));
  std::vector<wga_tok_err> errs(n_ok);
  d.download(errs.data(), d_err, n_ok);
  cb->d_ops = d_ops;
  cb->d_op_off = d_ooff;
  cb->d_strand_neg = d.upload(strand);
  cb->n_ops = total;
  if (all_errs) { /* the caller keeps the ops in front of a tokeniser error (call on PAF discards such errors) */
    *all_errs = errs;
    return first_err;
  }
  for (uint32_t k = 0; k < n_ok; k++)
    if (errs[k].err) { /* the batch is cut before the failing record: earlier ones may still fail in a walk */
      first_err = cigar_error_message(errs[k].err, cigars[k], (size_t)errs[k].tok_off, errs[k].tok_len);
      cb->n = k;
      std::vector<uint64_t> oo(1);
      d.download(oo.data(), d_ooff + k, 1);
      cb->n_ops = oo[0];
      break;
    }
  return first_err;
}

/* ---- rows of a batch of records -> MAF text (shared by paf2maf and chain2maf) ------------------------
 * The host supplies the fetched slices and the line text around the rows (MAFWriter, maf.rs:566-581:
 * "a score=..", "s\tname\tstart\tsize\tstrand\tsrcsize\t<row>" twice, blank line); K1, the layout scan,
 * K2 and the snippet scatter run on the device and one copy brings the finished text back. */
struct ExpandJob {
  std::vector<uint64_t> t_off, t_len, q_off, q_len;
  std::vector<uint32_t> pre_t, pre_q, post;
  std::string blob;
  std::vector<uint64_t> blob_off{0};
  void add(uint64_t to, uint64_t tl, uint64_t qo, uint64_t ql, uint64_t score, const std::string& t_name,
           uint64_t t_start, uint64_t t_ali, bool t_neg, uint64_t t_size, const std::string& q_name, uint64_t q_start,
           uint64_t q_ali, bool q_neg, uint64_t q_size) {
    t_off.push_back(to);
    t_len.push_back(tl);
    q_off.push_back(qo);
    q_len.push_back(ql);
    std::string a = "a score=";
    append_u64(a, score);
    a += "\ns\t" + t_name + "\t";
    append_u64(a, t_start);
    a.push_back('\t');
    append_u64(a, t_ali);
    a += t_neg ? "\t-\t" : "\t+\t";
    append_u64(a, t_size);
    a.push_back('\t');
    std::string q = "\ns\t" + q_name + "\t";
    append_u64(q, q_start);
    q.push_back('\t');
    append_u64(q, q_ali);
    q += q_neg ? "\t-\t" : "\t+\t";
    append_u64(q, q_size);
    q.push_back('\t');
    pre_t.push_back((uint32_t)a.size());
    pre_q.push_back((uint32_t)q.size());
    post.push_back(2);
    blob += a;
    blob_off.push_back(blob.size());
    blob += q;
    blob_off.push_back(blob.size());
    blob += "\n\n";
    blob_off.push_back(blob.size());
  }
  void resize(size_t k) {
    t_off.resize(k);
    t_len.resize(k);
    q_off.resize(k);
    q_len.resize(k);
    pre_t.resize(k);
    pre_q.resize(k);
    post.resize(k);
    blob_off.resize(3 * k + 1);
    blob.resize(blob_off.back());
  }
};

/* Where the text of a finished batch goes.  STREAM: into the command's output, in order (one device).  The other two serve
 * `--gpus N`, where a device holds every N-th target's records and the file is still written in input order: SIZES stops
 * after the layout scan and records every record's byte count (a record's size is known before a row byte exists), ROWS
 * writes each record with pwrite at the offset the sizes of all devices' records gave it. */
struct BatchSink {
  enum Mode { STREAM, SIZES, ROWS } mode = STREAM;
  Output* out = nullptr;
  std::vector<uint64_t>* sizes = nullptr;         /* SIZES: by record index of the piece */
  const std::vector<uint64_t>* offsets = nullptr; /* ROWS: file offset of every record of the piece */
  int fd = -1;
  const size_t* which = nullptr; /* record k of the batch is record which[first + k] of the piece */
  size_t first = 0;
};

/* n bytes of a device buffer to a file in runs: (offset in the buffer, length, file offset), through one pinned buffer */
void write_runs(Dev& d, int fd, const uint8_t* d_src, const std::vector<std::array<uint64_t, 3>>& runs) {
  static const size_t kPiece = (size_t)16 << 20;
  void* h = nullptr;
  d.check(wga_host_alloc(d.ctx, kPiece, &h));
  std::string err;
  for (const auto& r : runs) {
    for (uint64_t done = 0; done < r[1] && err.empty(); done += kPiece) {
      const size_t len = (size_t)std::min<uint64_t>(kPiece, r[1] - done);
      if (wga_memcpy_d2h_async(d.ctx, h, d_src + r[0] + done, len) || wga_sync(d.ctx)) {
        err = std::string("GPU engine: ") + wga_last_error();
        break;
      }
      size_t w = 0;
      while (w < len) {
        const ssize_t k = pwrite(fd, (const char*)h + w, len - w, (off_t)(r[2] + done + w));
        if (k <= 0) {
          err = "IO error:write failed";
          break;
        }
        w += (size_t)k;
      }
    }
    if (!err.empty()) break;
  }
  wga_host_free(d.ctx, h);
  if (!err.empty()) fail(err);
}

/* Writes (or sizes) the text of the leading records without a diagnostic; returns their number and, when it is
 * below cb.n, the diagnostic of the first failing record in *first_bad. */
uint32_t expand_batch(Dev& d, const wga_cigar_batch& cb, const ExpandJob& j, const uint8_t* d_tpool, uint64_t t_bytes,
                      const uint8_t* d_qpool, uint64_t q_bytes, BatchSink& sink, wga_rec_diag* first_bad) {
  const uint32_t n = cb.n;
  auto* d_counts = (wga_cigar_counts*)d.alloc((size_t)n * sizeof(wga_cigar_counts));
  auto* d_diag = (wga_rec_diag*)d.alloc((size_t)n * sizeof(wga_rec_diag));
  void* d_tiles = d.alloc(wga_tile_ws_bytes(cb.n_ops));
  d.check(wga_cigar_stat(d.ctx, &cb, d_counts, d_diag, d_tiles));
  uint64_t *d_to = d.upload(j.t_off), *d_tl = d.upload(j.t_len), *d_qo = d.upload(j.q_off), *d_ql = d.upload(j.q_len);
  uint32_t *d_pt = d.upload(j.pre_t), *d_pq = d.upload(j.pre_q), *d_po = d.upload(j.post);
  auto* d_tro = (uint64_t*)d.alloc((size_t)n * 8);
  auto* d_qro = (uint64_t*)d.alloc((size_t)n * 8);
  auto* d_rec = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
  d.check(wga_paf2maf_layout(d.ctx, n, d_counts, d_tl, d_ql, d_pt, d_pq, d_po, d_tro, d_qro, d_rec));
  std::vector<uint64_t> rec_off(n + 1), tro(n), qro(n);
  d.download(rec_off.data(), d_rec, n + 1);
  if (sink.mode == BatchSink::SIZES) { /* a record's byte count is known here, before a row byte exists */
    for (uint32_t k = 0; k < n; k++) (*sink.sizes)[sink.which ? sink.which[sink.first + k] : sink.first + k] = rec_off[k + 1] - rec_off[k];
    return n;
  }
  d.download(tro.data(), d_tro, n);
  d.download(qro.data(), d_qro, n);
  std::vector<wga_cigar_counts> counts(n);
  d.download(counts.data(), d_counts, n);
  d.out_arena_for(rec_off[n] + 64);
  d.check(wga_paf2maf_expand(d.ctx, &cb, d_counts, d_tiles, d_tpool, t_bytes, d_to, d_tl, d_qpool, q_bytes, d_qo, d_ql,
                             (uint8_t*)d.out_arena, d_tro, d_qro, d_diag));
  auto* d_out = (uint8_t*)d.out_arena;
  /* the MAF line text around the rows: three snippets per record */
  std::vector<uint64_t> dst(3 * (size_t)n);
  for (uint32_t k = 0; k < n; k++) {
    dst[3 * k] = rec_off[k];
    dst[3 * k + 1] = tro[k] + j.t_len[k] + counts[k].ins_bp + counts[k].inv_ins_bp;
    dst[3 * k + 2] = rec_off[k + 1] - 2;
  }
  uint8_t* d_blob = d.upload((const uint8_t*)j.blob.data(), j.blob.size());
  uint64_t *d_boff = d.upload(j.blob_off), *d_dst = d.upload(dst);
  d.check(wga_scatter_bytes(d.ctx, 3 * n, d_blob, d_boff, d_out, d_dst));
  std::vector<wga_rec_diag> diag(n);
  d.download(diag.data(), d_diag, n);
  uint32_t good = n;
  for (uint32_t k = 0; k < n; k++) {
    const wga_rec_diag& g = diag[k];
    if (g.bad_base_pos == WGA_NONE && g.bad_op_idx == WGA_NONE && g.panic_op_idx == WGA_NONE) continue;
    good = k;
    *first_bad = g;
    break;
  }
  if (sink.mode == BatchSink::STREAM) {
    stream_out(d, *sink.out, d_out, (size_t)rec_off[good]);
  } else { /* ROWS: every record where the input order puts it; neighbours in the input leave in one run */
    std::vector<std::array<uint64_t, 3>> runs;
    for (uint32_t k = 0; k < good; k++) {
      const uint64_t off = (*sink.offsets)[sink.which ? sink.which[sink.first + k] : sink.first + k], len = rec_off[k + 1] - rec_off[k];
      if (!runs.empty() && runs.back()[0] + runs.back()[1] == rec_off[k] && runs.back()[2] + runs.back()[1] == off)
        runs.back()[1] += len;
      else
        runs.push_back({rec_off[k], len, off});
    }
    d.check(wga_sync(d.ctx));
    write_runs(d, sink.fd, d_out, runs);
  }
  return good;
}

#include "cmd_paf2maf.inc"

#include "cmd_stat.inc"

#include "cmd_maf2paf.inc"

#include "cmd_validate.inc"

#include "cmd_chain.inc"

#include "cmd_dotplot.inc"

#include "cmd_pafcov.inc"

#include "cmd_pafpseudo.inc"

#include "cmd_call.inc"

/* ---- command line (cli.rs) -------------------------------------------------------------------------- */
void log_error(const std::string& msg) {
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  struct tm tmv;
  localtime_r(&ts.tv_sec, &tmv);
  char d[64], z[16];
  strftime(d, sizeof d, "%Y-%m-%dT%H:%M:%S", &tmv);
  strftime(z, sizeof z, "%z", &tmv);
  fprintf(stderr, "%s.%09ld%.3s:%.2s ERROR %s\n", d, ts.tv_nsec, z, z + 3, msg.c_str());
}

void usage() {
  fprintf(stderr,
          "wgatools (MI355X engine) — subcommands on the CIGAR hot path\n"
          "Usage: wgatools [-o OUT] [-r] [-t N] [-v] <COMMAND>\n"
          "  paf2maf | p2m  [PAF] -g TARGET.fa -q QUERY.fa\n"
          "  maf2paf | m2p  [MAF] [-q QUERY_NAME]\n"
          "  stat    | st   [FILE] [-f maf|paf] [-e] [-q QUERY_NAME]\n"
          "  pafcov  | pc   [PAF]\n"
          "  pafpseudo | pp [PAF] -o OUTDIR [-f ALL.fa] [-g TARGET]\n"
          "  validate | vf  [PAF] [-f FIXED.paf]\n"
          "  paf2chain | p2c [PAF]\n"
          "  maf2chain | m2c [MAF] [-q QUERY_NAME]\n"
          "  chain2paf | c2p [CHAIN]\n"
          "  dotplot | dp   [FILE] [-f maf|paf] --out-format csv [-m base-level|overview] [-d] [-l CUTOFF] [-q QUERY_NAME]\n"
          "  chain2maf | c2m [CHAIN] --target TARGET.fa --query QUERY.fa   (-g / -q)\n"
          "  call    | c    [MAF] [-s] [-i] [-l SVLEN] [-n SAMPLE] [--query-name N | --query-regex R] [-c CHUNK]\n"
          "  call    | c    -f paf [PAF] --target T.fa --query Q.fa [-s] [-l SVLEN] [-n SAMPLE]\n");
}

}  // namespace

static int run_command(int argc, char** argv) {
  try {
    std::string outfile = "-", cmd;
    bool rewrite = false;
    std::vector<std::string> rest;
    for (int i = 1; i < argc; i++) {
      std::string a = argv[i];
      auto val = [&](const char* name) -> std::string {
        if (i + 1 >= argc) fail(std::string("a value is required for '") + name + "'");
        return argv[++i];
      };
      if (a == "-o" || a == "--outfile")
        outfile = val("--outfile");
      else if (a.compare(0, 10, "--outfile=") == 0)
        outfile = a.substr(10);
      else if (a == "-r" || a == "--rewrite")
        rewrite = true;
      else if (a == "-t" || a == "--threads")
        (void)val("--threads");
      else if (a == "--spread")
        g_spread = true;
      else if (a == "--gpus") { /* ours: shard the PAF commands' records over N devices (paf2maf, stat -f paf, pafcov, pafpseudo, call -f paf) */
        const std::string v = val("--gpus");
        char* end = nullptr;
        const long n = strtol(v.c_str(), &end, 10);
        if (!end || *end || n < 1 || n > 64) fail("invalid value '" + v + "' for '--gpus <N>'");
        g_gpus = (int)n;
      }
      else if (a.size() >= 2 && a[0] == '-' && a.find_first_not_of('v', 1) == std::string::npos)
        ; /* -v, -vv, ... logging level */
      else if (a == "--verbose")
        ;
      else if (a == "-h" || a == "--help") {
        usage();
        return 0;
      } else if (cmd.empty() && a[0] != '-')
        cmd = a;
      else
        rest.push_back(a);
    }
    if (cmd.empty()) {
      usage();
      return 2;
    }
    if (g_gpus == 1) /* WGA_GPUS=N: the default of --gpus (a site-wide setting; the tests run whole suites under it) */
      if (const char* e = getenv("WGA_GPUS")) {
        const int n = atoi(e);
        if (n >= 1 && n <= 64) g_gpus = n;
      }
    if (g_gpus > 1 && g_gpus > wga_device_count())
      fail("--gpus " + std::to_string(g_gpus) + ": only " + std::to_string(wga_device_count()) + " device(s) visible");
    if (cmd.compare(0, 5, "__fmt") != 0) g_warm.start(); /* the HIP runtime comes up while the input is opened and read */
    /* hidden hooks for the CPU unit tests of the host formatters (no GPU involved) */
    if (cmd == "__fmt_f32") {
      for (const auto& a : rest) {
        uint32_t bits = (uint32_t)strtoul(a.c_str(), nullptr, 16);
        float f;
        memcpy(&f, &bits, 4);
        printf("%s\n", format_f32(f).c_str());
      }
      return 0;
    }
    if (cmd == "__fmt_f64") {
      for (const auto& a : rest) {
        uint64_t bits = strtoull(a.c_str(), nullptr, 16);
        double f;
        memcpy(&f, &bits, 8);
        printf("%s\n", format_f64(f).c_str());
      }
      return 0;
    }
    if (cmd == "__natord") {
      for (size_t i = 0; i + 1 < rest.size(); i += 2) printf("%d\n", natord_compare(rest[i], rest[i + 1]));
      return 0;
    }
    if (cmd == "__cs2cg") {
      for (const auto& a : rest) printf("%s\n", cs_to_cigar(a).c_str());
      return 0;
    }
    if (cmd == "__maf_reader") { /* which reader takes this file, and the blocks it yields */
      Dev d;
      MafInput min = load_maf(d, rest.empty() ? nullptr : &rest[0]);
      printf("%s\n%s\n", min.on_device ? "device" : "host", min.header.c_str());
      for (const auto& r : min.recs) {
        printf("block %zu", r.slines.size());
        for (const auto& sl : r.slines)
          printf(" [%s %llu %llu %c %llu %.*s]", sl.name.c_str(), (unsigned long long)sl.start,
                 (unsigned long long)sl.align_size, sl.neg ? '-' : '+', (unsigned long long)sl.size,
                 (int)std::min<size_t>(sl.seq_size(), 40), sl.seq_data());
        printf("\n");
      }
      return 0;
    }
    if (cmd == "__parse_chain") { /* host chain reader only: record and line counts */
      std::string e;
      std::vector<ChainRecord> recs = parse_chain(read_all(rest.empty() ? nullptr : &rest[0]), &e);
      size_t nl = 0;
      for (const auto& r : recs) nl += r.lines.size() / 3;
      printf("%zu records %zu lines %s\n", recs.size(), nl, e.c_str());
      return 0;
    }
    if (cmd == "__paf_reader") { /* which reader takes this file, and the fixed fields it yields */
      Dev d;
      PafInput pin = load_paf(d, rest.empty() ? nullptr : &rest[0], false);
      printf("%s\n", pin.on_device ? "device" : "host");
      for (size_t k = 0; k < pin.recs.size(); k++) {
        const PafRecord& r = pin.recs[k];
        printf("%s|%llu|%llu|%llu|%c|%s|%llu|%llu|%llu|%llu|%llu|%llu", r.query_name.c_str(),
               (unsigned long long)r.query_length, (unsigned long long)r.query_start,
               (unsigned long long)r.query_end, r.neg ? '-' : '+', r.target_name.c_str(),
               (unsigned long long)r.target_length, (unsigned long long)r.target_start,
               (unsigned long long)r.target_end, (unsigned long long)r.matches,
               (unsigned long long)r.block_length, (unsigned long long)r.mapq);
        if (pin.on_device) {
          if (pin.cg_beg[k] != WGA_NONE) printf("|cg:Z:%s", pin.text.substr(pin.cg_beg[k], pin.cg_end[k] - pin.cg_beg[k]).c_str());
        } else {
          int err = 0;
          std::string cg = paf_cigar_string(r, &err);
          if (!err) printf("|%s", cg.c_str());
        }
        printf("\n");
      }
      return 0;
    }
    if (cmd == "__parse_paf" || cmd == "__parse_maf") { /* echo the parsed records */
      std::string text = read_all(rest.empty() ? nullptr : &rest[0]);
      if (cmd == "__parse_paf") {
        for (const auto& r : parse_paf(text)) {
          printf("%s|%llu|%llu|%llu|%c|%s|%llu|%llu|%llu|%llu|%llu|%llu", r.query_name.c_str(),
                 (unsigned long long)r.query_length, (unsigned long long)r.query_start,
                 (unsigned long long)r.query_end, r.neg ? '-' : '+', r.target_name.c_str(),
                 (unsigned long long)r.target_length, (unsigned long long)r.target_start,
                 (unsigned long long)r.target_end, (unsigned long long)r.matches,
                 (unsigned long long)r.block_length, (unsigned long long)r.mapq);
          for (const auto& t : r.tags) printf("|%s", t.c_str());
          printf("\n");
        }
      } else {
        std::string header;
        for (const auto& r : parse_maf(text, &header)) {
          printf("block %zu", r.slines.size());
          for (const auto& sl : r.slines)
            printf(" [%s %llu %llu %c %llu %zu]", sl.name.c_str(), (unsigned long long)sl.start,
                   (unsigned long long)sl.align_size, sl.neg ? '-' : '+', (unsigned long long)sl.size,
                   sl.seq_size());
          printf("\n");
        }
      }
      return 0;
    }
    /* per-subcommand options */
    std::string input_s, target, query, format = "maf", query_name, fasta, sample = "sample", query_regex;
    bool has_input = false, each = false, has_qname = false, has_fasta = false, snp = false, inv = false,
         has_regex = false;
    uint64_t svlen = 50, chunk_size = 1000000;
    const bool pseudo = cmd == "pafpseudo" || cmd == "pp";
    const bool call = cmd == "call" || cmd == "c";
    const bool validate = cmd == "validate" || cmd == "vf";
    const bool dotp = cmd == "dotplot" || cmd == "dp";
    std::string out_format = "html", mode = "base-level";
    bool no_identity = false, has_cutoff = false;
    uint64_t cutoff = 50; /* utils.rs:709-710 */
    std::string fix_path;
    bool has_fix = false;
    for (size_t i = 0; i < rest.size(); i++) {
      const std::string& a = rest[i];
      auto val = [&]() -> std::string {
        if (i + 1 >= rest.size()) fail("a value is required for '" + a + "'");
        return rest[++i];
      };
      bool conv = cmd == "paf2maf" || cmd == "p2m" || cmd == "chain2maf" || cmd == "c2m";
      if (a == "-g" || a == "--target")
        target = val();
      else if ((a == "-q" || a == "--query") && (conv || call))
        query = val();
      else if (a == "-q" || a == "--query-name") {
        query_name = val();
        has_qname = true;
      } else if ((a == "-f" || a == "--fix") && validate) {
        fix_path = val();
        has_fix = true;
      } else if ((a == "-f" || a == "--fasta") && pseudo) {
        fasta = val();
        has_fasta = true;
      } else if (a == "-f" || a == "--format")
        format = val();
      else if (dotp && a == "--out-format")
        out_format = val();
      else if (dotp && (a == "-m" || a == "--mode"))
        mode = val();
      else if (dotp && (a == "-d" || a == "--no-identity"))
        no_identity = true;
      else if (dotp && (a == "-l" || a == "--length")) {
        cutoff = strtoull(val().c_str(), nullptr, 10);
        has_cutoff = true;
      } else if (dotp && a == "--color")
        (void)val(); /* colours only exist in the Vega-Lite outputs */
      else if (a == "-e" || a == "--each")
        each = true;
      else if (call && (a == "-s" || a == "--snp"))
        snp = true;
      else if (call && (a == "-i" || a == "--inv"))
        inv = true;
      else if (call && (a == "-l" || a == "--svlen"))
        svlen = strtoull(val().c_str(), nullptr, 10);
      else if (call && a.size() > 2 && a.compare(0, 2, "-l") == 0)
        svlen = strtoull(a.c_str() + 2, nullptr, 10);
      else if (call && (a == "-n" || a == "--sample"))
        sample = val();
      else if (call && (a == "-c" || a == "--chunk-size"))
        chunk_size = strtoull(val().c_str(), nullptr, 10);
      else if (call && a == "--query-regex") {
        query_regex = val();
        has_regex = true;
      }
      else if (a[0] != '-' && !has_input) {
        input_s = a;
        has_input = true;
      } else
        fail("unexpected argument '" + a + "'");
    }
    const std::string* input = has_input ? &input_s : nullptr;
    const std::string* qn = has_qname ? &query_name : nullptr;
    Output out;
    if (cmd == "paf2maf" || cmd == "p2m") {
      if (target.empty() || query.empty()) fail("the following required arguments were not provided: --target --query");
      out.open(outfile, rewrite);
      return cmd_paf2maf(input, target, query, out);
    }
    if (cmd == "stat" || cmd == "st") {
      out.open(outfile, rewrite);
      if (format == "paf") return cmd_stat_paf(input, each, out);
      if (format == "maf") return cmd_stat_maf(input, each, qn, out);
      fail("format `" + format + "` is not supported by this engine (maf | paf)");
    }
    if (cmd == "paf2chain" || cmd == "p2c") {
      out.open(outfile, rewrite);
      return cmd_paf2chain(input, out);
    }
    if (dotp) {
      (void)has_cutoff;
      out.open(outfile, rewrite);
      return cmd_dotplot(input, format, out_format, mode, no_identity, cutoff, qn, out);
    }
    if (cmd == "maf2chain" || cmd == "m2c") {
      out.open(outfile, rewrite);
      return cmd_maf2chain(input, qn, out);
    }
    if (cmd == "chain2paf" || cmd == "c2p") {
      out.open(outfile, rewrite);
      return cmd_chain2paf(input, out);
    }
    if (cmd == "chain2maf" || cmd == "c2m") {
      if (target.empty() || query.empty()) fail("the following required arguments were not provided: --target --query");
      out.open(outfile, rewrite);
      return cmd_chain2maf(input, target, query, out);
    }
    if (validate) {
      if (has_fix && fix_path == (has_input ? input_s : std::string("stdin")))
        fail("fixed file should not be the same as output file"); /* utils.rs:754-758 */
      out.open(outfile, rewrite);
      return cmd_validate(input, has_fix ? &fix_path : nullptr, out);
    }
    if (cmd == "maf2paf" || cmd == "m2p") {
      out.open(outfile, rewrite);
      return cmd_maf2paf(input, qn, out);
    }
    if (call) {
      if (format == "paf") {
        if (target.empty() || query.empty()) fail("target and query are necessary"); /* main.rs:103-110 */
        out.open(outfile, rewrite);
        return cmd_call_paf(input, target, query, snp, svlen, sample, out);
      }
      if (format != "maf") fail("format is not supported");
      out.open(outfile, rewrite);
      return cmd_call_maf(input, snp, inv, svlen, sample, qn, has_regex ? &query_regex : nullptr, chunk_size, out);
    }
    if (pseudo)
      return cmd_pafpseudo(input, outfile, rewrite, has_fasta ? &fasta : nullptr, target.empty() ? nullptr : &target);
    if (cmd == "pafcov" || cmd == "pc") {
      out.open(outfile, rewrite);
      return cmd_pafcov(input, out, g_spread);
    }
    fail("subcommand `" + cmd + "` is not on the CIGAR hot path and is not provided by this engine");
  } catch (Error& e) {
    log_error(e.msg);
    return 1;
  }
  return 0;
}

int main(int argc, char** argv) { return run_command(argc, argv); }
