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
  void run(Output& out, const uint8_t* d_src, size_t n) {
    if (n == 0) return;
    const size_t np = (n + kPiece - 1) / kPiece;
    /* writers: memcpy into the page cache is what bounds a large output (one thread moves 2-3 GB/s), the copy engine delivers
     * far more; WGA_WRITE_THREADS overrides */
    int nthreads = 1;
    {
      uint64_t pos_probe = 0;
      if (out.plain_fd(&pos_probe) >= 0) {
        nthreads = 8; /* 16 / 24 / 32 measured slower at 15 GB (scripts/gpu_write_threads.py: 2.6 / 2.9 / 3.0 s against 2.5) */
        if (const char* e = getenv("WGA_WRITE_THREADS")) nthreads = std::max(1, std::min(32, atoi(e)));
      }
    }
    const int nbuf = (int)std::min<size_t>((size_t)std::min(kBufs, nthreads + 4), np + 1); /* pinned memory is slow to get: only what this run uses */
    for (int k = 0; k < nbuf; k++)
      if (!buf[k] && wga_host_alloc(ctx, kPiece, &buf[k])) fail(std::string("GPU engine: ") + wga_last_error());
    uint64_t pos0 = 0;
    const int fd = out.plain_fd(&pos0);
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

struct Dev {
  wga_ctx* ctx = nullptr;
  bool own_ctx = true; /* false: the context is another Dev's (the reader's, lent to device 0's worker) */
  int device = 0;
  std::vector<void*> owned;
  std::unique_ptr<DevStreamer> streamer; /* pinned buffers, allocated once per process */
  Dev() {}
  explicit Dev(int dev) : device(dev) {}
  void init() {
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
    if (ctx) {
      streamer.reset();
      if (out_arena) wga_free(ctx, out_arena);
      for (void* p : owned) wga_free(ctx, p);
      if (own_ctx) wga_ctx_destroy(ctx);
    }
  }
};
static void stream_out(Dev& d, Output& out, const uint8_t* d_src, size_t n) {
  if (!d.streamer) d.streamer.reset(new DevStreamer(d.ctx));
  d.check(wga_sync(d.ctx));
  g_timer.mark("kernels + host tables");
  d.streamer->run(out, d_src, n);
  g_timer.mark("copy out + write");
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

/* ---- paf2maf (converter.rs:176-265) ------------------------------------------------------------- */
/* one thread per device over fn(g); the first worker error (by device number) is rethrown on the caller's thread */
static void on_devices(int ngpu, const std::function<void(int)>& fn) {
  std::vector<std::string> werr(ngpu);
  std::vector<std::thread> th;
  for (int g = 0; g < ngpu; g++)
    th.emplace_back([&, g] {
      try {
        fn(g);
      } catch (Error& e) {
        werr[g] = e.msg.empty() ? std::string("error") : e.msg;
      } catch (std::exception& e) {
        werr[g] = std::string("internal error: ") + e.what();
      }
    });
  for (auto& t : th) t.join();
  for (int g = 0; g < ngpu; g++)
    if (!werr[g].empty()) fail(werr[g]);
}

/* fnv1a64(name): the sharding rule of the multi-device paths (the same hash as wgatools_amd/shard.py) */
static uint64_t fnv1a64(const std::string& s) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (unsigned char c : s) h = (h ^ c) * 0x100000001B3ull;
  return h;
}

/* The body of the converter's record loop over records which[0 .. n_which) of a piece (which == nullptr: all of them), in
 * resident batches: slices fetched (target first, then query: converter.rs:219-225), CIGARs tokenised on the device, rows
 * expanded, text handed to the sink.  Returns the position (in `which` order) of the first failing record, or n_which;
 * `err` = the reference's message for it. */
size_t p2m_run(Dev& d, DevFasta& tf, DevFasta& qf, const PafInput& in, const size_t* which, size_t n_which,
               const uint8_t* d_text_here, BatchSink sink, std::string& err) {
  const std::vector<PafRecord>& recs = in.recs;
  auto rec_of = [&](size_t k) -> const PafRecord& { return recs[which ? which[k] : k]; };
  const uint64_t kMaxBytes = 6ull << 30;
  const uint64_t kMaxText = 160ull << 20; /* ~64 M ops */
  const size_t keep = d.owned.size();
  size_t i0 = 0;
  while (i0 < n_which && err.empty()) {
    ExpandJob job;
    uint64_t est = 0, est_text = 0;
    size_t i = i0;
    for (; i < n_which; i++) {
      const PafRecord& r = rec_of(i);
      if (i > i0 && (est_text > kMaxText || est > kMaxBytes)) break;
      uint64_t to, tl, qo, ql;
      try { /* fetch order of converter.rs:219-225: target first, then query */
        tf.fetch(r.target_name, r.target_start, r.target_end - 1, &to, &tl);
        qf.fetch(r.query_name, r.query_start, r.query_end - 1, &qo, &ql);
      } catch (Error& e) {
        err = e.msg;
        break;
      }
      est_text += in.cigar_bytes(which ? which[i] : i);
      job.add(to, tl, qo, ql, r.mapq, r.target_name, r.target_start, r.target_end - r.target_start, false,
              r.target_length, r.query_name, r.neg ? r.query_length - r.query_end : r.query_start, /* converter.rs:213-216 */
              r.query_end - r.query_start, r.neg, r.query_length);
      est += tl + ql + (tl + ql) / 4;
    }
    size_t bad_at = err.empty() ? n_which : i; /* a fetch error belongs to record i */
    /* the CIGARs of records [i0, i) are tokenised on the device; a tag / tokeniser error cuts the
     * batch before the failing record (reverse_complement runs before the CIGAR is looked at, so
     * an invalid base in that record's query slice still wins: checked on the host, rare path) */
    CigarTexts cigars;
    wga_cigar_batch cb;
    cb.n = 0;
    if (i > i0) {
      const std::string terr = device_tokenise(d, in, i0, (uint32_t)(i - i0), cigars, &cb, nullptr, which, d_text_here);
      if (!terr.empty()) {
        const size_t k = cb.n;
        std::string perr = terr;
        const PafRecord& r = rec_of(i0 + k);
        if (r.neg) {
          const std::string qs = qf.slice(d, job.q_off[k], job.q_len[k]);
          for (uint64_t x = job.q_len[k]; x-- > 0;) {
            char c = qs[x];
            if (!strchr("ACGTNacgtn", c) || c == 0) {
              perr = std::string("Invalid Base: `") + c + "`";
              break;
            }
          }
        }
        err = perr;
        i = i0 + k;
        bad_at = i;
      }
    }
    const uint32_t n = cb.n;
    if (n) {
      job.resize(n);
      wga_rec_diag g;
      sink.which = which;
      sink.first = i0;
      const uint32_t good = expand_batch(d, cb, job, tf.d_pool, tf.bytes, qf.d_pool, qf.bytes, sink, &g);
      if (good < n) {
        const uint32_t k = good;
        if (g.bad_base_pos != WGA_NONE) { /* utils.rs:97 */
          char c = qf.at(d, job.q_off[k] + job.q_len[k] - 1 - g.bad_base_pos);
          err = std::string("Invalid Base: `") + c + "`";
        } else if (g.bad_op_idx < g.panic_op_idx) { /* errors.rs:59 */
          err = "CIGAR OP `" + cigar_op_token_at(cigars[k], g.bad_op_idx) + "` invalid";
        } else {
          err = "panic: String::insert_str beyond the end of the fetched sequence (cigar.rs:507,513)";
        }
        bad_at = i0 + k;
      }
    }
    d.release_to(keep); /* this batch's buffers */
    if (!err.empty()) return bad_at;
    i0 = i;
  }
  return n_which;
}

/* `wgatools --gpus N paf2maf`: a piece of the PAF is framed once (device 0 splits it), every record belongs to device
 * fnv1a64(target_name) % N, and N worker threads — one context, one pair of sequence pools each — run the same record loop
 * over their share: a first pass for the byte count of every record (K1 + the layout scan; no row byte yet), then, with the
 * file offsets those sizes give in INPUT order, a second pass that expands the rows and pwrite()s every record where it
 * belongs.  No row byte crosses devices; the first failing record in input order ends the run and the file ends in front
 * of it, as the reference's serial loop leaves it (converter.rs:196-263). */
int cmd_paf2maf_multi(const std::string* input, const std::string& t_fa, const std::string& q_fa, Output& out, int ngpu) {
  std::vector<std::unique_ptr<Dev>> devs;
  for (int g = 0; g < ngpu; g++) devs.emplace_back(new Dev(g));
  PafChunks chunks(input, false);
  std::vector<DevFasta> tf(ngpu), qf(ngpu);
  auto on_all = [&](const std::function<void(int)>& fn) { on_devices(ngpu, fn); };
  on_all([&](int g) {
    devs[g]->init();
    tf[g].load(*devs[g], t_fa);
    qf[g].load(*devs[g], q_fa);
  });
  g_timer.mark("fasta read + device pools");
  std::vector<size_t> keep_pools(ngpu);
  for (int g = 0; g < ngpu; g++) keep_pools[g] = devs[g]->owned.size(); /* the pools stay for the whole run */
  out.write("#maf version=1.6 convert_from=paf t_seq_path=" + t_fa + " q_seq_path=" + q_fa + "\n");
  uint64_t pos0 = 0;
  const int fd = out.plain_fd(&pos0);
  if (fd < 0) fail("internal error: --gpus needs a plain output file");
  uint64_t file_pos = pos0;
  std::string pending_error;
  PafInput in;
  for (;;) {
    bool more = false;
    try {
      more = chunks.next(*devs[0], in);
    } catch (Error& e) {
      pending_error = e.msg;
    }
    g_timer.mark("paf read + upload + split");
    if (!more) break;
    const size_t n = in.recs.size();
    std::vector<std::vector<size_t>> mine(ngpu);
    for (size_t i = 0; i < n; i++) mine[fnv1a64(in.recs[i].target_name) % (uint64_t)ngpu].push_back(i);
    std::vector<const uint8_t*> d_text(ngpu, nullptr);
    std::vector<uint64_t> sizes(n, 0), offsets(n + 1, 0);
    std::vector<size_t> bad_at(ngpu, n); /* input index of a worker's first failing record */
    std::vector<std::string> bad_msg(ngpu);
    std::string text16;
    if (in.on_device && ngpu > 1) text16 = in.text + std::string(16, '\0');
    auto pass = [&](BatchSink::Mode mode, size_t upto) {
      on_all([&](int g) {
        Dev& d = *devs[g];
        if (in.on_device && !d_text[g])
          d_text[g] = g == 0 ? in.d_text : d.upload((const uint8_t*)text16.data(), text16.size());
        std::vector<size_t>& w = mine[g];
        size_t cnt = std::lower_bound(w.begin(), w.end(), upto) - w.begin(); /* records in front of the first known error */
        BatchSink sink;
        sink.mode = mode;
        sink.sizes = &sizes;
        sink.offsets = &offsets;
        sink.fd = fd;
        std::string err;
        const size_t k = p2m_run(d, tf[g], qf[g], in, w.data(), cnt, d_text[g], sink, err);
        if (k < cnt && w[k] < bad_at[g]) {
          bad_at[g] = w[k];
          bad_msg[g] = err;
        }
      });
    };
    pass(BatchSink::SIZES, n);
    size_t first_bad = n;
    for (int g = 0; g < ngpu; g++) first_bad = std::min(first_bad, bad_at[g]);
    offsets[0] = file_pos;
    for (size_t i = 0; i < n; i++) offsets[i + 1] = offsets[i] + (i < first_bad ? sizes[i] : 0);
    g_timer.mark("sizes (K1 + layout on every device)");
    pass(BatchSink::ROWS, first_bad);
    for (int g = 0; g < ngpu; g++) first_bad = std::min(first_bad, bad_at[g]);
    g_timer.mark("rows + copy out + write");
    file_pos = offsets[first_bad];
    for (int g = 0; g < ngpu; g++) devs[g]->release_to(keep_pools[g]); /* the piece's text on every device */
    if (first_bad < n) {
      for (int g = 0; g < ngpu; g++)
        if (bad_at[g] == first_bad) pending_error = bad_msg[g];
      break;
    }
  }
  /* the file ends behind the last record in front of the first failing one (later records of other devices may have
   * been written beyond it) */
  if (ftruncate(fd, (off_t)file_pos) != 0) fail("IO error:truncate failed");
  out.advance(file_pos - pos0);
  out.close();
  if (!pending_error.empty()) fail(pending_error);
  return 0;
}

int cmd_paf2maf(const std::string* input, const std::string& t_fa, const std::string& q_fa, Output& out) {
  {
    uint64_t pos = 0;
    if (g_gpus > 1 && out.plain_fd(&pos) >= 0) return cmd_paf2maf_multi(input, t_fa, q_fa, out, g_gpus);
  }
  Dev d;
  PafChunks chunks(input, false); /* the input is opened first, then the two indexed FASTA files (utils.rs, converter.rs:183-186) */
  DevFasta tf, qf;
  d.init();
  tf.load(d, t_fa);
  qf.load(d, q_fa);
  g_timer.mark("fasta read + device pools");
  out.write("#maf version=1.6 convert_from=paf t_seq_path=" + t_fa + " q_seq_path=" + q_fa + "\n");
  const size_t keep_pools = d.owned.size(); /* the pools stay for the whole run */
  std::string pending_error;
  PafInput in;
  /* the input streams through in pieces (records before a failing one are written, like the reference's reader loop) */
  for (;;) {
    bool more = false;
    g_timer.mark("host");
    try {
      more = chunks.next(d, in);
    } catch (Error& e) {
      pending_error = e.msg;
    }
    g_timer.mark("paf read + upload + split");
    if (!more) break;
    BatchSink sink;
    sink.out = &out;
    p2m_run(d, tf, qf, in, nullptr, in.recs.size(), nullptr, sink, pending_error);
    d.release_to(keep_pools);
    if (!pending_error.empty()) break;
  }
  out.close();
  if (!pending_error.empty()) fail(pending_error);
  return 0;
}

/* ---- stat (stat.rs) ----------------------------------------------------------------------------- */
/* `wgatools --gpus N stat -f paf`: the records of a piece are counted on device fnv1a64(target_name) % N; the per-record
 * counters meet on the host, where the Pair group-by runs as on one device (stat.rs:167-223).  No collective: what a
 * reduction over devices would sum — the grand totals — is the last line of the host's merge. */
static void stat_piece_multi(std::vector<std::unique_ptr<Dev>>& devs, const PafInput& pin, std::vector<wga_cigar_counts>& counts) {
  const int ngpu = (int)devs.size();
  const size_t n = pin.recs.size();
  std::vector<std::vector<size_t>> mine(ngpu);
  for (size_t i = 0; i < n; i++) mine[fnv1a64(pin.recs[i].target_name) % (uint64_t)ngpu].push_back(i);
  std::vector<size_t> bad_at(ngpu, n);
  std::vector<std::string> bad_msg(ngpu);
  std::string text16;
  if (pin.on_device && ngpu > 1) text16 = pin.text + std::string(16, '\0');
  on_devices(ngpu, [&](int g) {
    Dev& d = *devs[g];
    d.init();
    const std::vector<size_t>& w = mine[g];
    if (w.empty()) return;
    const size_t keep = d.owned.size();
    const uint8_t* d_text = nullptr;
    if (pin.on_device) d_text = g == 0 ? pin.d_text : d.upload((const uint8_t*)text16.data(), text16.size());
    CigarTexts cigars;
    wga_cigar_batch cb;
    const std::string e = device_tokenise(d, pin, 0, (uint32_t)w.size(), cigars, &cb, nullptr, w.data(), d_text);
    const uint32_t m = cb.n; /* records before this device's first tag / tokeniser error */
    if (m) {
      auto* d_counts = (wga_cigar_counts*)d.alloc((size_t)m * sizeof(wga_cigar_counts));
      auto* d_diag = (wga_rec_diag*)d.alloc((size_t)m * sizeof(wga_rec_diag));
      d.check(wga_cigar_stat(d.ctx, &cb, d_counts, d_diag, nullptr));
      std::vector<wga_rec_diag> diag(m);
      std::vector<wga_cigar_counts> c(m);
      d.download(diag.data(), d_diag, m);
      d.download(c.data(), d_counts, m);
      for (uint32_t k = 0; k < m; k++) {
        if (diag[k].bad_op_idx != WGA_NONE) {
          bad_at[g] = w[k];
          bad_msg[g] = "CIGAR OP `" + cigar_op_token_at(cigars[k], diag[k].bad_op_idx) + "` invalid";
          break;
        }
        counts[w[k]] = c[k];
      }
    }
    if (!e.empty() && w[m] < bad_at[g]) {
      bad_at[g] = w[m];
      bad_msg[g] = e;
    }
    d.release_to(keep); /* this piece's buffers (the reader's own copy of the text on device 0 is its to release) */
  });
  size_t first_bad = n;
  for (int g = 0; g < ngpu; g++) first_bad = std::min(first_bad, bad_at[g]);
  if (first_bad < n) /* buffered driver: nothing is written on error; the first failing record in input order speaks */
    for (int g = 0; g < ngpu; g++)
      if (bad_at[g] == first_bad) fail(bad_msg[g]);
}

int cmd_stat_paf(const std::string* input, bool each, Output& out) {
  Dev d;
  PafChunks chunks(input, false);
  std::vector<StatInput> in;
  PafInput pin;
  std::vector<std::unique_ptr<Dev>> devs; /* --gpus N: devices 1 .. N - 1 next to `d` */
  while (chunks.next(d, pin)) { /* one piece of the file at a time; only the per-record statistics are kept */
    const std::vector<PafRecord>& recs = pin.recs;
    const uint32_t n = (uint32_t)recs.size();
    std::vector<wga_cigar_counts> counts(n);
    d.init();
    if (g_gpus > 1) {
      if (devs.empty()) {
        devs.emplace_back(new Dev(0));
        devs[0]->ctx = d.ctx; /* device 0's context is the reader's */
        devs[0]->own_ctx = false;
        for (int g = 1; g < g_gpus; g++) devs.emplace_back(new Dev(g));
      }
      stat_piece_multi(devs, pin, counts);
      in.reserve(in.size() + n);
      for (uint32_t k = 0; k < n; k++) {
        const PafRecord& r = recs[k];
        in.push_back(StatInput{r.target_name, r.query_name, r.target_length, r.query_length, r.target_start,
                               r.query_start, recstat_from(counts[k])});
      }
      d.release_all();
      continue;
    }
    CigarTexts cigars;
    wga_cigar_batch cb;
    const std::string e = device_tokenise(d, pin, 0, n, cigars, &cb);
    const uint32_t m = cb.n; /* records before the first tag / tokeniser error */
    if (m) {
      auto* d_counts = (wga_cigar_counts*)d.alloc((size_t)m * sizeof(wga_cigar_counts));
      auto* d_diag = (wga_rec_diag*)d.alloc((size_t)m * sizeof(wga_rec_diag));
      d.check(wga_cigar_stat(d.ctx, &cb, d_counts, d_diag, nullptr));
      std::vector<wga_rec_diag> diag(m);
      d.download(diag.data(), d_diag, m);
      d.download(counts.data(), d_counts, m);
      for (uint32_t k = 0; k < m; k++)
        if (diag[k].bad_op_idx != WGA_NONE)
          fail("CIGAR OP `" + cigar_op_token_at(cigars[k], diag[k].bad_op_idx) + "` invalid");
    }
    if (!e.empty()) fail(e); /* buffered driver: nothing is written on error */
    in.reserve(in.size() + n);
    for (uint32_t k = 0; k < n; k++) {
      const PafRecord& r = recs[k];
      in.push_back(StatInput{r.target_name, r.query_name, r.target_length, r.query_length, r.target_start,
                             r.query_start, recstat_from(counts[k])});
    }
    d.release_all();
  }
  out.write(stat_tsv(in, each));
  out.close();
  return 0;
}

/* A MAF input: blocks with their s-lines.  Plain files are split on the device (wga_maf_split): the rows
 * are never copied — the K3 / K4 walks read them in the uploaded file, the host keeps spans into its own
 * copy of the text for the few places that need row characters (VCF REF / ALT).  A file with a line the
 * splitter does not take goes through the host reader (the reference's errors). */
struct MafInput {
  std::shared_ptr<std::string> text = std::make_shared<std::string>();
  std::string header;
  std::vector<MafRecord> recs;
  bool on_device = false;
  uint8_t* d_text = nullptr;
};
MafInput maf_from_text(Dev& d, std::string&& whole_text) {
  MafInput in;
  *in.text = std::move(whole_text);
  const std::string& text = *in.text;
  const char* force = getenv("WGA_MAF_READER"); /* "host": always the host reader (measurements) */
  if (!text.empty() && text.size() < 0xFFFFFFF0ull && !(force && strcmp(force, "host") == 0)) {
    g_timer.mark("file read");
    d.init();
    in.text->append(16, '\0'); /* slack behind the text for whole-vector loads */
    in.d_text = d.upload((const uint8_t*)text.data(), text.size());
    in.text->resize(text.size() - 16);
    g_timer.mark("upload");
    uint64_t n_lines = 0;
    d.check(wga_maf_split(d.ctx, in.d_text, text.size(), &n_lines, nullptr, 0));
    auto* d_lines = (wga_maf_line*)d.alloc((size_t)(n_lines + 1) * sizeof(wga_maf_line));
    d.check(wga_maf_split(d.ctx, in.d_text, text.size(), &n_lines, d_lines, n_lines));
    std::vector<wga_maf_line> lines((size_t)n_lines);
    if (n_lines) d.download(lines.data(), d_lines, (size_t)n_lines);
    d.release(d_lines);
    g_timer.mark("device split + line table");
    bool plain = true;
    for (const wga_maf_line& L : lines)
      if (L.status == WGA_MAF_FALLBACK) plain = false;
    if (plain) {
      in.on_device = true;
      size_t he = text.find('\n'); /* the first line is always the header (maf.rs:25-36) */
      if (he == std::string::npos) he = text.size();
      if (he > 0 && text[he - 1] == '\r') he--;
      in.header.assign(text, 0, he);
      /* a block = a maximal run of s-lines (any other line ends the block in progress); the records of a piece with millions of
       * blocks are filled by a few threads (two or more strings per block: the allocator is what this loop costs) */
      std::vector<std::pair<size_t, size_t>> runs_of_s; /* [first, behind the last) s-line of every block */
      for (size_t i = 0; i < lines.size();) {
        if (lines[i].status != WGA_MAF_SLINE) {
          i++;
          continue;
        }
        size_t j = i;
        while (j < lines.size() && lines[j].status == WGA_MAF_SLINE) j++;
        runs_of_s.emplace_back(i, j);
        i = j;
      }
      in.recs.resize(runs_of_s.size());
      auto fill = [&](size_t b0, size_t b1) {
        for (size_t b = b0; b < b1; b++) {
          MafRecord& r = in.recs[b];
          r.slines.reserve(runs_of_s[b].second - runs_of_s[b].first);
          for (size_t i = runs_of_s[b].first; i < runs_of_s[b].second; i++) {
            const wga_maf_line& L = lines[i];
            MafSLine sl;
            sl.name.assign(text, (size_t)L.name_off, L.name_len);
            sl.start = L.num[0];
            sl.align_size = L.num[1];
            sl.size = L.num[2];
            sl.neg = L.strand_neg != 0;
            sl.file = text.data();
            sl.seq_off = L.seq_off;
            sl.seq_len = L.seq_len;
            r.slines.push_back(std::move(sl));
          }
        }
      };
      const size_t nb = runs_of_s.size();
      static const size_t min_blocks = getenv("WGA_MAF_FILL_MIN_BLOCKS") ? (size_t)strtoull(getenv("WGA_MAF_FILL_MIN_BLOCKS"), nullptr, 10) : 50000; /* tests: 1 */
      const unsigned T = nb >= std::max<size_t>(min_blocks, 8) ? 8u : 1u;
      if (T == 1u) {
        fill(0, nb);
      } else {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++) th.emplace_back(fill, nb * t / T, nb * (t + 1) / T);
        fill(0, nb / T);
        for (auto& x : th) x.join();
      }
      g_timer.mark("host records");
      return in;
    }
    d.release(in.d_text);
    in.d_text = nullptr;
  }
  in.recs = parse_maf(text, &in.header);
  return in;
}
MafInput load_maf(Dev& d, const std::string* input) { return maf_from_text(d, read_all(input)); }

/* A MAF input in pieces of about 1 GiB (WGA_CHUNK_BYTES) that end between blocks: a piece is cut in front of its trailing
 * run of `s` lines, which open the next piece.  The reader's rule that the file's first line is the header is kept for
 * the later pieces by a dummy first line. */
struct MafChunks {
  LineChunkReader rd;
  size_t target = (size_t)1 << 30;
  std::string pending; /* trailing s-lines of the previous piece */
  bool first = true, done = false;
  std::string header;
  std::unique_ptr<BgzfDeviceSource> bgzf; /* a bgzipped input: inflated on the device */
  explicit MafChunks(const std::string* input) {
    bgzf.reset(new BgzfDeviceSource());
    if (bgzf->open(input))
      rd.source = [this](char* dst, size_t want) { return bgzf->read(dst, want); };
    else
      bgzf.reset();
    rd.open(input);
    if (const char* e = getenv("WGA_CHUNK_BYTES")) target = (size_t)strtoull(e, nullptr, 10);
    if (target == 0) target = 1;
  }
  /* reader side: the text of the next piece that holds something, or false at the end of the input */
  bool produce(std::string& text) {
    while (!done) {
      text = first ? std::string() : std::string("#\n");
      const size_t skip = text.size();
      text += pending;
      pending.clear();
      const bool more = rd.next(text, target, text.size()); /* reads behind the prefix, no second copy */
      if (!more) {
        done = true;
      } else { /* cut in front of the trailing run of lines that start with 's' (the header line never counts) */
        size_t cut = text.size();
        while (cut > skip) {
          size_t ls = cut >= 2 ? text.rfind('\n', cut - 2) : std::string::npos; /* start of the last line in [.., cut) */
          ls = (ls == std::string::npos || ls + 1 < skip) ? skip : ls + 1;
          const bool is_header = first && ls == 0;
          if (text[ls] == 's' && !is_header)
            cut = ls;
          else
            break;
        }
        if (cut == skip && text.size() > skip) { /* nothing but s-lines so far: one block longer than a piece */
          pending.assign(text, skip, std::string::npos);
          continue;
        }
        pending.assign(text, cut, std::string::npos);
        text.resize(cut);
      }
      if (text.size() == skip) continue;
      first = false;
      return true;
    }
    return false;
  }
  /* the next piece is read by a helper thread while the caller works on the current one (as PafChunks does) */
  struct Ahead {
    bool ok = false;
    std::string text, err;
  } ahead;
  std::thread reader;
  bool started = false, first_seen = true;
  ~MafChunks() {
    if (reader.joinable()) reader.join();
  }
  void read_ahead() {
    reader = std::thread([this] {
      Ahead a;
      try {
        a.ok = produce(a.text);
      } catch (Error& e) {
        a.err = e.msg.empty() ? std::string("error") : e.msg;
      } catch (std::exception& e) {
        a.err = std::string("internal error: ") + e.what();
      }
      ahead = std::move(a);
    });
  }
  /* the next piece with at least one block, or false at the end of the input */
  bool next(Dev& d, MafInput& in) {
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
      if (in.text && in.text.use_count() == 1) rd.recycle(std::move(*in.text)); /* nobody else holds the piece the caller is done with */
      read_ahead();
      in = maf_from_text(d, std::move(a.text));
      if (first_seen) header = in.header;
      first_seen = false;
      if (!in.recs.empty()) return true;
      if (in.d_text) d.release(in.d_text);
    }
  }
};

/* the (target row, query row) pairs of a list of blocks on the device: offsets into the uploaded file, or
 * — host reader — into one buffer the rows are gathered in */
struct MafRows {
  const uint8_t* d_rows = nullptr;
  uint64_t *d_t = nullptr, *d_q = nullptr, *d_c = nullptr;
  uint8_t* d_s = nullptr;
  std::vector<uint64_t> cols;
};
MafRows device_rows(Dev& d, const MafInput& in, const std::vector<const MafRecord*>& recs, bool cols_target) {
  MafRows m;
  std::vector<uint64_t> t_off, q_off;
  std::vector<uint8_t> strand;
  std::string blob;
  for (const MafRecord* r : recs) {
    const MafSLine &t = r->t(), &q = r->q();
    if (in.on_device) {
      t_off.push_back(t.seq_off);
      q_off.push_back(q.seq_off);
    } else {
      t_off.push_back(blob.size());
      blob.append(t.seq_data(), t.seq_size());
      q_off.push_back(blob.size());
      blob.append(q.seq_data(), q.seq_size());
    }
    /* zip truncates to the shorter row; `call` walks the target row's length (caller.rs:115) */
    m.cols.push_back(cols_target ? t.seq_size() : std::min(t.seq_size(), q.seq_size()));
    strand.push_back(q.neg ? 1 : 0);
  }
  d.init();
  m.d_rows = in.on_device ? in.d_text : d.upload((const uint8_t*)blob.data(), blob.size());
  m.d_t = d.upload(t_off);
  m.d_q = d.upload(q_off);
  m.d_c = d.upload(m.cols);
  m.d_s = d.upload(strand);
  return m;
}
std::vector<const MafRecord*> all_records(const std::vector<MafRecord>& recs) {
  std::vector<const MafRecord*> v;
  v.reserve(recs.size());
  for (const auto& r : recs) v.push_back(&r);
  return v;
}
/* --gpus N for the MAF commands: blocks are independent, so the selected blocks of a piece are dealt out in N contiguous
 * ranges; device 0 reads the rows where the piece was uploaded, the others get a gathered copy of their range's rows.
 * fn(g, dev, rows, lo, count) runs on one thread per device; results are merged by the caller in block order. */
struct MafDevices {
  Dev& d0;
  std::vector<std::unique_ptr<Dev>> extra;
  explicit MafDevices(Dev& first) : d0(first) {
    for (int g = 1; g < g_gpus; g++) extra.emplace_back(new Dev(g));
  }
  int count() const { return 1 + (int)extra.size(); }
  Dev& dev(int g) { return g == 0 ? d0 : *extra[g - 1]; }
  void release_all() {
    if (d0.ctx) d0.release_all();
    for (auto& e : extra)
      if (e->ctx) e->release_all(); /* a device that got no block of this piece was never started */
  }
  void run(const MafInput& in, const std::vector<const MafRecord*>& recs, bool cols_target,
           const std::function<void(int, Dev&, const MafRows&, uint32_t, uint32_t)>& fn) {
    const uint32_t n = (uint32_t)recs.size();
    const int ng = count();
    if (ng == 1) {
      MafRows p = device_rows(d0, in, recs, cols_target);
      fn(0, d0, p, 0, n);
      return;
    }
    on_devices(ng, [&](int g) {
      const uint32_t lo = (uint32_t)((uint64_t)n * g / ng), hi = (uint32_t)((uint64_t)n * (g + 1) / ng);
      if (lo == hi) return;
      std::vector<const MafRecord*> part(recs.begin() + lo, recs.begin() + hi);
      MafInput host_view; /* rows gathered from the host copy of the text */
      MafRows p = device_rows(dev(g), g == 0 ? in : host_view, part, cols_target);
      fn(g, dev(g), p, lo, hi - lo);
    });
  }
};

void select_query(std::vector<MafRecord>& recs, const std::string* query_name) {
  for (auto& r : recs) {
    if (query_name) { /* maf.rs:277-285 */
      size_t k = 0;
      for (; k < r.slines.size(); k++)
        if (r.slines[k].name == *query_name) break;
      if (k == r.slines.size()) fail("Query name:" + *query_name + " not found in MAF");
      r.query_idx = k;
    }
    if (r.query_idx >= r.slines.size())
      fail("panic: MAF block with a single s-line has no query row (maf.rs:426 index out of bounds)");
  }
}

int cmd_stat_maf(const std::string* input, bool each, const std::string* query_name, Output& out) {
  Dev d;
  MafDevices md(d);
  MafChunks chunks(input);
  std::vector<StatInput> in;
  MafInput min;
  while (chunks.next(d, min)) { /* one piece of the file at a time; only the per-block statistics are kept */
    std::vector<MafRecord>& recs = min.recs;
    select_query(recs, query_name);
    const uint32_t n = (uint32_t)recs.size();
    std::vector<wga_cigar_counts> counts(n);
    md.run(min, all_records(recs), false, [&](int, Dev& dg, const MafRows& p, uint32_t lo, uint32_t cnt) {
      auto* d_counts = (wga_cigar_counts*)dg.alloc((size_t)cnt * sizeof(wga_cigar_counts));
      auto* d_cnt = (uint64_t*)dg.alloc((size_t)cnt * 8);
      dg.check(wga_maf_pair_stat(dg.ctx, cnt, p.d_rows, p.d_t, p.d_q, p.d_c, p.d_s, d_counts, d_cnt, nullptr, nullptr));
      dg.download(counts.data() + lo, d_counts, cnt);
    });
    for (uint32_t k = 0; k < n; k++) {
      const MafRecord& r = recs[k];
      in.push_back(StatInput{r.t().name, r.q().name, r.t().size, r.q().size, r.t().start, r.query_start(),
                             recstat_from(counts[k])});
    }
    md.release_all();
  }
  out.write(stat_tsv(in, each));
  out.close();
  return 0;
}

/* ---- maf2paf (converter.rs:29-54, maf.rs:484-520) ------------------------------------------------ */
/* the PAF rows of blocks recs[0 .. n), whose rows stand on device d (p) */
static std::string maf2paf_rows(Dev& d, const MafRows& p, const MafRecord* const* recs, uint32_t n) {
  std::string text;
  const uint8_t* d_rows = p.d_rows;
  auto *d_t = p.d_t, *d_q = p.d_q, *d_c = p.d_c;
  auto* d_s = p.d_s;
  auto* d_counts = (wga_cigar_counts*)d.alloc((size_t)n * sizeof(wga_cigar_counts));
  auto* d_cnt = (uint64_t*)d.alloc((size_t)n * 8);
  d.check(wga_maf_pair_stat(d.ctx, n, d_rows, d_t, d_q, d_c, d_s, d_counts, d_cnt, nullptr, nullptr));
  auto* d_roff = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
  d.check(wga_exclusive_scan_u64(d.ctx, n, d_cnt, d_roff));
  std::vector<uint64_t> roff(n + 1);
  d.download(roff.data(), d_roff, n + 1);
  auto* d_runs = (uint64_t*)d.alloc((roff[n] + 1) * 8);
  d.check(wga_maf_pair_stat(d.ctx, n, d_rows, d_t, d_q, d_c, d_s, d_counts, d_cnt, d_runs, d_roff));
  /* the cg:Z: text is formatted on the device (wga_maf_runs_cigar_text) between the host's fields */
  auto* d_tb = (uint64_t*)d.alloc((size_t)n * 8);
  d.check(wga_maf_runs_cigar_text(d.ctx, n, roff[n], d_runs, d_roff, d_c, d_tb, nullptr, nullptr));
  std::vector<uint64_t> tb(n);
  d.download(tb.data(), d_tb, n);
  std::vector<wga_cigar_counts> counts(n);
  d.download(counts.data(), d_counts, n);
  std::string blob;
  std::vector<uint64_t> blob_off{0}, dst, text_off(n);
  uint64_t pos = 0;
  for (uint32_t k = 0; k < n; k++) {
    const MafRecord& r = *recs[k];
    const wga_cigar_counts& c = counts[k];
    uint64_t block = c.match + c.mismatch + c.ins_bp + c.inv_ins_bp + c.del_bp + c.inv_del_bp;
    std::string h;
    append_csv_field(h, r.q().name, '\t');
    uint64_t a[] = {r.q().size, r.query_start(), r.query_end()};
    for (uint64_t v : a) {
      h.push_back('\t');
      append_u64(h, v);
    }
    h += r.q().neg ? "\t-\t" : "\t+\t";
    append_csv_field(h, r.t().name, '\t');
    uint64_t bb[] = {r.t().size, r.t().start, r.t().start + r.t().align_size, c.match, block, 255};
    for (uint64_t v : bb) {
      h.push_back('\t');
      append_u64(h, v);
    }
    h += "\tNM:i:";
    append_u64(h, block - c.match);
    h += "\tcg:Z:";
    dst.push_back(pos);
    blob += h;
    blob_off.push_back(blob.size());
    pos += h.size();
    text_off[k] = pos;
    pos += tb[k];
    dst.push_back(pos);
    blob += "\n";
    blob_off.push_back(blob.size());
    pos += 1;
  }
  auto* d_out = (uint8_t*)d.alloc(pos + 64);
  d.check(wga_maf_runs_cigar_text(d.ctx, n, roff[n], d_runs, d_roff, d_c, nullptr, d_out, d.upload(text_off)));
  d.check(wga_scatter_bytes(d.ctx, 2 * n, d.upload((const uint8_t*)blob.data(), blob.size()), d.upload(blob_off), d_out,
                            d.upload(dst)));
  text.resize((size_t)pos);
  if (pos) d.download((uint8_t*)&text[0], d_out, pos);
  return text;
}

int cmd_maf2paf(const std::string* input, const std::string* query_name, Output& out) {
  Dev d;
  MafDevices md(d); /* --gpus N: a piece's blocks in contiguous ranges over the devices, the rows meet in block order */
  MafChunks chunks(input);
  std::string all_text; /* converter.rs:40-52 collects every record before it writes the first: an error leaves no output */
  MafInput min;
  while (chunks.next(d, min)) {
    std::vector<MafRecord>& recs = min.recs;
    select_query(recs, query_name);
    if (!recs.empty()) {
      const std::vector<const MafRecord*> all = all_records(recs);
      std::vector<std::string> part(md.count());
      md.run(min, all, false, [&](int g, Dev& dg, const MafRows& p, uint32_t lo, uint32_t cnt) {
        part[g] = maf2paf_rows(dg, p, all.data() + lo, cnt);
      });
      for (const std::string& t : part) all_text += t;
    }
    md.release_all();
  }
  out.write(all_text);
  out.close();
  return 0;
}

/* ---- validate (validate.rs:44-141; SURVEY.md 8f rank 3: free once K1 exists) ------------------------------
 * query_start + M + X + I must be query_end, target_start + M + X + D must be target_end.  Report to the
 * output, optionally all records with corrected ends to --fix.  Lists are in input order (the reference's
 * par_bridge order is not deterministic). */
int cmd_validate(const std::string* input, const std::string* fix, Output& out) {
  Dev d;
  PafChunks chunks(input, fix != nullptr); /* --fix re-serialises every tag: host reader */
  uint64_t n_total = 0, q_bad = 0, t_bad = 0;
  std::string q_list, t_list, rows;
  PafInput pin;
  std::vector<std::unique_ptr<Dev>> devs; /* --gpus N: devices 1 .. N - 1 next to `d` */
  while (chunks.next(d, pin)) { /* one piece of the file at a time */
    std::vector<PafRecord>& recs = pin.recs;
    const uint32_t n = (uint32_t)recs.size();
    std::vector<wga_cigar_counts> counts(n);
    d.init();
    std::string e;
    if (g_gpus > 1) { /* --gpus N: the records by target hash, as `stat` (the counts meet on the host) */
      if (devs.empty()) {
        devs.emplace_back(new Dev(0));
        devs[0]->ctx = d.ctx; /* device 0's context is the reader's */
        devs[0]->own_ctx = false;
        for (int g = 1; g < g_gpus; g++) devs.emplace_back(new Dev(g));
      }
      try {
        stat_piece_multi(devs, pin, counts);
      } catch (Error& er) {
        e = er.msg;
      }
    } else {
      CigarTexts cigars;
      wga_cigar_batch cb;
      e = device_tokenise(d, pin, 0, n, cigars, &cb);
      const uint32_t m = cb.n;
      if (m) {
        auto* d_counts = (wga_cigar_counts*)d.alloc((size_t)m * sizeof(wga_cigar_counts));
        auto* d_diag = (wga_rec_diag*)d.alloc((size_t)m * sizeof(wga_rec_diag));
        d.check(wga_cigar_stat(d.ctx, &cb, d_counts, d_diag, nullptr));
        std::vector<wga_rec_diag> diag(m);
        d.download(diag.data(), d_diag, m);
        d.download(counts.data(), d_counts, m);
        for (uint32_t k = 0; k < m && e.empty(); k++)
          if (diag[k].bad_op_idx != WGA_NONE) {
            e = "CIGAR OP `" + cigar_op_token_at(cigars[k], diag[k].bad_op_idx) + "` invalid";
            break;
          }
      }
    }
    /* rec.get_stat().unwrap() (:79) */
    if (!e.empty()) fail("panic: called `Result::unwrap()` on an `Err` value: " + e);
    n_total += n;
    for (uint32_t k = 0; k < n; k++) {
      PafRecord& r = recs[k];
      const wga_cigar_counts& c = counts[k];
      const uint64_t mx = c.match + c.mismatch;
      const uint64_t eq = r.query_start + mx + c.ins_bp + c.inv_ins_bp, et = r.target_start + mx + c.del_bp + c.inv_del_bp;
      if (eq != r.query_end) {
        q_bad++;
        q_list += r.query_name + ":";
        append_u64(q_list, r.query_start);
        q_list.push_back('-');
        append_u64(q_list, r.query_end);
        q_list.push_back('\n');
        r.query_end = eq;
      }
      if (et != r.target_end) {
        t_bad++;
        t_list += r.target_name + ":";
        append_u64(t_list, r.target_start);
        t_list.push_back('-');
        append_u64(t_list, r.target_end);
        t_list.push_back('\n');
        r.target_end = et;
      }
    }
    if (fix) /* csv writer: tab, flexible, no header; PafRecord field order (paf.rs:50-65) */
      for (const PafRecord& r : recs) {
        append_csv_field(rows, r.query_name, '\t');
        const uint64_t a[] = {r.query_length, r.query_start, r.query_end};
        for (uint64_t v : a) {
          rows.push_back('\t');
          append_u64(rows, v);
        }
        rows += r.neg ? "\t-\t" : "\t+\t";
        append_csv_field(rows, r.target_name, '\t');
        const uint64_t b2[] = {r.target_length, r.target_start, r.target_end, r.matches, r.block_length, r.mapq};
        for (uint64_t v : b2) {
          rows.push_back('\t');
          append_u64(rows, v);
        }
        for (const std::string& tg : r.tags) {
          rows.push_back('\t');
          append_csv_field(rows, tg, '\t');
        }
        rows.push_back('\n');
      }
    d.release_all();
  }
  std::string text = "Total records: ";
  append_u64(text, n_total);
  text += "\nQuery invalid records: ";
  append_u64(text, q_bad);
  text += "\nTarget invalid records: ";
  append_u64(text, t_bad);
  text += "\nQuery invalid list:\n" + q_list + "Target invalid list:\n" + t_list + "\n"; /* writeln!("{}", ..) */
  out.write(text);
  if (fix) {
    if (*fix == "-") {
      out.write(rows);
    } else {
      Output fo;
      fo.open(*fix, true);
      fo.write(rows);
      fo.close();
    }
  }
  out.close();
  return 0;
}

/* ---- paf2chain (converter.rs:148-173) ------------------------------------------------
 * GPU: tokeniser, data lines and head / tail trims (wga_cigar_chain).  Host: chain headers
 * (chain.rs:142-203, incl. the '-' strand arithmetic that reuses the updated start) and the layout. */
/* the chains of records [lo, hi) of a piece on device d, in resident batches; a batch's text goes to `sink` while it is still on
 * the device.  Returns the reference's message for the first failing record ("" if none): the records in front of it are written. */
static std::string paf2chain_range(Dev& d, const PafInput& pin, size_t lo, size_t hi, uint64_t chain_base,
                                   const uint8_t* d_text_here,
                                   const std::function<void(Dev&, const uint8_t*, size_t)>& sink) {
  const std::vector<PafRecord>& recs = pin.recs;
  const uint64_t kMaxText = 160ull << 20;
  std::string pending_error;
  const size_t keep = d.owned.size(); /* this piece's text */
  size_t i0 = lo;
  while (i0 < hi && pending_error.empty()) {
    size_t i = i0;
    uint64_t est_text = 0;
    for (; i < hi; i++) {
      if (i > i0 && est_text > kMaxText) break;
      est_text += pin.cigar_bytes(i);
    }
    d.init();
    CigarTexts cigars;
    wga_cigar_batch cb;
    pending_error = device_tokenise(d, pin, i0, (uint32_t)(i - i0), cigars, &cb, nullptr, nullptr, d_text_here);
    uint32_t n = cb.n;
    if (n) {
      auto* d_trim = (wga_chain_trim_t*)d.alloc((size_t)n * sizeof(wga_chain_trim_t));
      auto* d_nb = (uint64_t*)d.alloc((size_t)n * 8);
      auto* d_diag = (wga_rec_diag*)d.alloc((size_t)n * sizeof(wga_rec_diag));
      d.check(wga_cigar_chain(d.ctx, &cb, d_trim, d_nb, d_diag, nullptr, nullptr));
      std::vector<wga_chain_trim_t> trim(n);
      std::vector<uint64_t> nb(n);
      std::vector<wga_rec_diag> diag(n);
      d.download(trim.data(), d_trim, n);
      d.download(nb.data(), d_nb, n);
      d.download(diag.data(), d_diag, n);
      for (uint32_t k = 0; k < n; k++)
        if (diag[k].bad_op_idx != WGA_NONE) { /* parse_cigar_to_trim fails before anything of the record is written */
          pending_error = "CIGAR OP `" + cigar_op_token_at(cigars[k], diag[k].bad_op_idx) + "` invalid";
          n = k;
          break;
        }
      if (n) {
        std::string blob;
        std::vector<uint64_t> blob_off{0}, dst, data_off(n);
        uint64_t pos = 0;
        for (uint32_t k = 0; k < n; k++) {
          const PafRecord& r = recs[i0 + k];
          const wga_chain_trim_t& t = trim[k];
          uint64_t qs = r.query_start, qe = r.query_end, ts = r.target_start + t.head_del, te = r.target_end - t.tail_del;
          if (!r.neg) {
            qs += t.head_ins;
            qe -= t.tail_ins;
          } else {
            qs = r.query_length - (qe - t.head_ins);
            qe = r.query_length - (qs + t.tail_ins);
          }
          std::string h = "chain\t255\t" + r.target_name + "\t";
          append_u64(h, r.target_length);
          h += "\t+\t";
          append_u64(h, ts);
          h.push_back('\t');
          append_u64(h, te);
          h += "\t" + r.query_name + "\t";
          append_u64(h, r.query_length);
          h += r.neg ? "\t-\t" : "\t+\t";
          append_u64(h, qs);
          h.push_back('\t');
          append_u64(h, qe);
          h.push_back('\t');
          append_u64(h, chain_base + (uint64_t)(i0 + k));
          dst.push_back(pos);
          blob += h;
          blob_off.push_back(blob.size());
          pos += h.size();
          data_off[k] = pos;
          pos += nb[k];
          dst.push_back(pos);
          blob += "\n\n";
          blob_off.push_back(blob.size());
          pos += 2;
        }
        auto* d_out = (uint8_t*)d.alloc(pos + 64);
        wga_cigar_batch cb2 = cb;
        cb2.n = n;
        d.check(wga_cigar_chain(d.ctx, &cb2, nullptr, nullptr, nullptr, d_out, d.upload(data_off)));
        d.check(wga_scatter_bytes(d.ctx, 2 * n, d.upload((const uint8_t*)blob.data(), blob.size()), d.upload(blob_off),
                                  d_out, d.upload(dst)));
        sink(d, d_out, (size_t)pos);
      }
    }
    d.release_to(keep);
    i0 = i;
  }
  return pending_error;
}

int cmd_paf2chain(const std::string* input, Output& out) {
  Dev d;
  PafChunks chunks(input, false);
  std::string pending_error;
  PafInput pin;
  uint64_t chain_base = 0; /* chain id = index of the record in the whole input */
  std::vector<std::unique_ptr<Dev>> devs; /* --gpus N: devices 1 .. N - 1 next to `d` */
  for (;;) {
    bool more = false;
    try {
      more = chunks.next(d, pin);
    } catch (Error& e) {
      pending_error = e.msg;
    }
    if (!more) break;
    const size_t n = pin.recs.size();
    if (g_gpus == 1) {
      pending_error = paf2chain_range(d, pin, 0, n, chain_base, nullptr,
                                      [&](Dev& dg, const uint8_t* d_out, size_t bytes) { stream_out(dg, out, d_out, bytes); });
    } else { /* a piece's records in contiguous ranges over the devices; the chains meet in input order, up to the first error */
      d.init();
      if (devs.empty()) {
        devs.emplace_back(new Dev(0));
        devs[0]->ctx = d.ctx; /* device 0's context is the reader's */
        devs[0]->own_ctx = false;
        for (int g = 1; g < g_gpus; g++) devs.emplace_back(new Dev(g));
      }
      const int ng = g_gpus;
      std::vector<std::string> part(ng), err(ng);
      std::string text16;
      if (pin.on_device) text16 = pin.text + std::string(16, '\0');
      on_devices(ng, [&](int g) {
        const size_t lo = n * (size_t)g / ng, hi = n * (size_t)(g + 1) / ng;
        if (lo == hi) return;
        Dev& dg = *devs[g];
        dg.init();
        const size_t keep = dg.owned.size();
        const uint8_t* d_text = nullptr;
        if (pin.on_device) d_text = g == 0 ? pin.d_text : dg.upload((const uint8_t*)text16.data(), text16.size());
        err[g] = paf2chain_range(dg, pin, lo, hi, chain_base, d_text, [&](Dev& dd, const uint8_t* d_out, size_t bytes) {
          const size_t at = part[g].size();
          part[g].resize(at + bytes);
          if (bytes) dd.download((uint8_t*)&part[g][at], d_out, bytes);
        });
        dg.release_to(keep);
      });
      for (int g = 0; g < ng && pending_error.empty(); g++) {
        out.write(part[g]);
        pending_error = err[g];
      }
    }
    chain_base += n;
    d.release_all();
    if (!pending_error.empty()) break;
  }
  out.close();
  if (!pending_error.empty()) fail(pending_error);
  return 0;
}

/* ---- the chain readers' device batch: data lines -> packed ops (wga_chain_lines_ops) ----------------- */
struct ChainBatch {
  wga_cigar_batch cb;
  uint64_t* d_lines = nullptr;
  uint64_t* d_line_off = nullptr;
  uint64_t n_lines = 0;
};
ChainBatch chain_device_batch(Dev& d, const ChainRecord* recs, uint32_t n) {
  ChainBatch b;
  std::vector<uint64_t> lines, line_off{0};
  std::vector<uint8_t> strand;
  for (uint32_t k = 0; k < n; k++) {
    lines.insert(lines.end(), recs[k].lines.begin(), recs[k].lines.end());
    line_off.push_back(lines.size() / 3);
    strand.push_back(recs[k].query_neg ? 1 : 0);
  }
  b.n_lines = lines.size() / 3;
  lines.resize(lines.size() + 3);
  b.d_lines = d.upload(lines);
  b.d_line_off = d.upload(line_off);
  auto* d_cnt = (uint64_t*)d.alloc((size_t)n * 8);
  d.check(wga_chain_lines_ops(d.ctx, n, b.n_lines, b.d_lines, b.d_line_off, d_cnt, nullptr, nullptr));
  auto* d_ooff = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
  d.check(wga_exclusive_scan_u64(d.ctx, n, d_cnt, d_ooff));
  uint64_t total = 0;
  d.download(&total, d_ooff + n, 1);
  auto* d_ops = (uint32_t*)d.alloc((total + 4) * 4);
  d.check(wga_chain_lines_ops(d.ctx, n, b.n_lines, b.d_lines, b.d_line_off, nullptr, d_ops, d_ooff));
  b.cb.d_ops = d_ops;
  b.cb.d_op_off = d_ooff;
  b.cb.d_strand_neg = d.upload(strand);
  b.cb.n_ops = total;
  b.cb.n = n;
  return b;
}

/* ---- chain2maf (converter.rs:268-358) ------------------------------------------------------------------
 * A data line (size, dt, dq) is the op group "size M, dq I, dt D" of parse_chain_to_insert (:360-388), so the
 * rows come from the same kernels as paf2maf.  Records before a failing one are written, like the reference. */
/* the converter's record loop over chains [lo, hi), in resident batches: slices fetched (:309-316: target first, then query, both on
 * the forward strand), data lines -> ops, rows expanded, text handed to the sink (record k of the run is record k of the file).
 * Returns the index of the first failing record, or hi; `err` = the reference's message for it. */
static size_t c2m_run(Dev& d, DevFasta& tf, DevFasta& qf, const std::vector<ChainRecord>& recs, size_t lo, size_t hi, BatchSink sink,
                      std::string& err) {
  const uint64_t kMaxBytes = 6ull << 30, kMaxLines = 32ull << 20;
  const size_t keep = d.owned.size(); /* the pools stay */
  size_t i0 = lo;
  while (i0 < hi) {
    ExpandJob job;
    uint64_t est = 0, est_lines = 0;
    size_t i = i0;
    std::string fetch_error;
    for (; i < hi; i++) {
      const ChainRecord& r = recs[i];
      if (i > i0 && (est > kMaxBytes || est_lines > kMaxLines)) break;
      uint64_t to, tl, qo, ql;
      try {
        tf.fetch(r.target_name, r.target_start, r.target_end - 1, &to, &tl);
        qf.fetch(r.query_name, r.query_start, r.query_end - 1, &qo, &ql);
      } catch (Error& e) {
        fetch_error = e.msg;
        break;
      }
      job.add(to, tl, qo, ql, 255, r.target_name, r.target_start, r.target_end - r.target_start, r.target_neg,
              r.target_size, r.query_name, r.query_neg ? r.query_size - r.query_end : r.query_start, /* :299-302 */
              r.query_end - r.query_start, r.query_neg, r.query_size);
      est += tl + ql + (tl + ql) / 4;
      est_lines += r.lines.size() / 3;
    }
    const uint32_t n = (uint32_t)(i - i0);
    if (n) {
      ChainBatch b = chain_device_batch(d, &recs[i0], n);
      wga_rec_diag g;
      sink.which = nullptr;
      sink.first = i0;
      const uint32_t good = expand_batch(d, b.cb, job, tf.d_pool, tf.bytes, qf.d_pool, qf.bytes, sink, &g);
      d.check(wga_sync(d.ctx));
      if (good < n) {
        if (g.bad_base_pos != WGA_NONE) { /* utils.rs:97, reverse_complement of the query slice (:320-325) */
          char c = qf.at(d, job.q_off[good] + job.q_len[good] - 1 - g.bad_base_pos);
          err = std::string("Invalid Base: `") + c + "`";
        } else {
          err = "panic: String::insert_str beyond the end of the fetched sequence (converter.rs:375,383)";
        }
        d.release_to(keep);
        return i0 + good;
      }
      d.release_to(keep);
    }
    if (!fetch_error.empty()) {
      err = fetch_error;
      return i;
    }
    i0 = i;
  }
  return hi;
}

/* `wgatools --gpus N chain2maf`: the chains in N contiguous ranges, both pools on every device; sizes first (K1 + layout), a
 * prefix gives every record its file offset, rows second, every device pwrite()s its records; the file ends in front of the
 * first failing record. */
static int cmd_chain2maf_multi(std::vector<ChainRecord>& recs, std::string pending_error, const std::string& t_fa,
                               const std::string& q_fa, Output& out, int ngpu) {
  std::vector<std::unique_ptr<Dev>> devs;
  for (int g = 0; g < ngpu; g++) devs.emplace_back(new Dev(g));
  std::vector<DevFasta> tf(ngpu), qf(ngpu);
  on_devices(ngpu, [&](int g) {
    devs[g]->init();
    tf[g].load(*devs[g], t_fa);
    qf[g].load(*devs[g], q_fa);
  });
  out.write("#maf version=1.6 convert_from=chain t_seq_path=" + t_fa + " q_seq_path=" + q_fa + "\n");
  uint64_t pos0 = 0;
  const int fd = out.plain_fd(&pos0);
  if (fd < 0) fail("internal error: --gpus needs a plain output file");
  const size_t n = recs.size();
  std::vector<uint64_t> sizes(n, 0), offsets(n + 1, 0);
  std::vector<size_t> bad_at(ngpu, n);
  std::vector<std::string> bad_msg(ngpu);
  auto pass = [&](BatchSink::Mode mode, size_t upto) {
    on_devices(ngpu, [&](int g) {
      const size_t lo = n * (size_t)g / ngpu, hi = std::min(upto, n * (size_t)(g + 1) / ngpu);
      if (lo >= hi) return;
      BatchSink sink;
      sink.mode = mode;
      sink.sizes = &sizes;
      sink.offsets = &offsets;
      sink.fd = fd;
      std::string err;
      const size_t k = c2m_run(*devs[g], tf[g], qf[g], recs, lo, hi, sink, err);
      if (k < hi && k < bad_at[g]) {
        bad_at[g] = k;
        bad_msg[g] = err;
      }
    });
  };
  pass(BatchSink::SIZES, n);
  size_t first_bad = n;
  for (int g = 0; g < ngpu; g++) first_bad = std::min(first_bad, bad_at[g]);
  offsets[0] = pos0;
  for (size_t i = 0; i < n; i++) offsets[i + 1] = offsets[i] + (i < first_bad ? sizes[i] : 0);
  pass(BatchSink::ROWS, first_bad);
  for (int g = 0; g < ngpu; g++) first_bad = std::min(first_bad, bad_at[g]);
  if (first_bad < n)
    for (int g = 0; g < ngpu; g++)
      if (bad_at[g] == first_bad) pending_error = bad_msg[g];
  if (ftruncate(fd, (off_t)offsets[first_bad]) != 0) fail("IO error:truncate failed");
  out.advance(offsets[first_bad] - pos0);
  out.close();
  if (!pending_error.empty()) fail(pending_error);
  return 0;
}

int cmd_chain2maf(const std::string* input, const std::string& t_fa, const std::string& q_fa, Output& out) {
  std::string pending_error;
  std::vector<ChainRecord> recs = parse_chain(read_all(input), &pending_error);
  {
    uint64_t pos = 0;
    if (g_gpus > 1 && recs.size() > 1 && out.plain_fd(&pos) >= 0)
      return cmd_chain2maf_multi(recs, pending_error, t_fa, q_fa, out, g_gpus);
  }
  DevFasta tf, qf;
  Dev d;
  d.init();
  tf.load(d, t_fa);
  qf.load(d, q_fa);
  out.write("#maf version=1.6 convert_from=chain t_seq_path=" + t_fa + " q_seq_path=" + q_fa + "\n");
  BatchSink sink;
  sink.out = &out;
  std::string err;
  if (c2m_run(d, tf, qf, recs, 0, recs.size(), sink, err) < recs.size()) pending_error = err;
  out.close();
  if (!pending_error.empty()) fail(pending_error);
  return 0;
}

/* ---- chain2paf (converter.rs:391-416, chain.rs:430-452) ------------------------------------------------
 * GPU: data lines -> ops -> K1 (matches, block length) and the CIGAR text (wga_chain_lines_cigar_text).
 * All records are converted before the first is written (:402-410): an error leaves the output empty. */
/* the PAF rows of chains recs[0 .. n_recs) on device d, in resident batches; a batch's text goes to `sink` while it is on the device */
static void chain2paf_range(Dev& d, const ChainRecord* recs, size_t n_recs,
                            const std::function<void(Dev&, const uint8_t*, size_t)>& sink) {
  d.init();
  const size_t keep = d.owned.size();
  const uint64_t kMaxLines = 32ull << 20;
  size_t i0 = 0;
  while (i0 < n_recs) {
    size_t i = i0;
    uint64_t est = 0;
    for (; i < n_recs; i++) {
      if (i > i0 && est > kMaxLines) break;
      est += recs[i].lines.size() / 3;
    }
    d.init();
    const uint32_t n = (uint32_t)(i - i0);
    ChainBatch b = chain_device_batch(d, &recs[i0], n);
    auto* d_counts = (wga_cigar_counts*)d.alloc((size_t)n * sizeof(wga_cigar_counts));
    auto* d_diag = (wga_rec_diag*)d.alloc((size_t)n * sizeof(wga_rec_diag));
    d.check(wga_cigar_stat(d.ctx, &b.cb, d_counts, d_diag, nullptr));
    std::vector<wga_cigar_counts> counts(n);
    d.download(counts.data(), d_counts, n);
    auto* d_tb = (uint64_t*)d.alloc((size_t)n * 8);
    d.check(wga_chain_lines_cigar_text(d.ctx, n, b.n_lines, b.d_lines, b.d_line_off, d_tb, nullptr, nullptr));
    std::vector<uint64_t> tb(n);
    d.download(tb.data(), d_tb, n);
    /* record text = host fields up to "cg:Z:" | device CIGAR | "\n" */
    std::string blob;
    std::vector<uint64_t> blob_off{0}, dst, text_off(n);
    uint64_t pos = 0;
    for (uint32_t k = 0; k < n; k++) {
      const ChainRecord& r = recs[i0 + k];
      const wga_cigar_counts& c = counts[k];
      std::string h;
      append_csv_field(h, r.query_name, '\t');
      const uint64_t a[] = {r.query_size, r.query_start, r.query_end};
      for (uint64_t v : a) {
        h.push_back('\t');
        append_u64(h, v);
      }
      h += r.query_neg ? "\t-\t" : "\t+\t";
      append_csv_field(h, r.target_name, '\t');
      /* block_length = match + mismatch + del + inv_del (chain.rs:433-435) */
      const uint64_t b2[] = {r.target_size, r.target_start, r.target_end, c.match,
                             c.match + c.mismatch + c.del_bp + c.inv_del_bp, 255};
      for (uint64_t v : b2) {
        h.push_back('\t');
        append_u64(h, v);
      }
      h += "\tcg:Z:";
      dst.push_back(pos);
      blob += h;
      blob_off.push_back(blob.size());
      pos += h.size();
      text_off[k] = pos;
      pos += tb[k];
      dst.push_back(pos);
      blob += "\n";
      blob_off.push_back(blob.size());
      pos += 1;
    }
    auto* d_out = (uint8_t*)d.alloc(pos + 64);
    d.check(wga_chain_lines_cigar_text(d.ctx, n, b.n_lines, b.d_lines, b.d_line_off, nullptr, d_out, d.upload(text_off)));
    d.check(wga_scatter_bytes(d.ctx, 2 * n, d.upload((const uint8_t*)blob.data(), blob.size()), d.upload(blob_off), d_out,
                              d.upload(dst)));
    sink(d, d_out, (size_t)pos);
    d.release_to(keep);
    i0 = i;
  }
}

int cmd_chain2paf(const std::string* input, Output& out) {
  std::string perr;
  std::vector<ChainRecord> recs = parse_chain(read_all(input), &perr);
  if (!perr.empty()) {
    out.close();
    fail(perr);
  }
  Dev d;
  if (g_gpus == 1 || recs.size() < 2) {
    if (!recs.empty())
      chain2paf_range(d, recs.data(), recs.size(),
                      [&](Dev& dg, const uint8_t* d_out, size_t bytes) { stream_out(dg, out, d_out, bytes); });
  } else { /* --gpus N: the chains in contiguous ranges over the devices, the rows meet in input order */
    const int ng = g_gpus;
    const size_t n = recs.size();
    std::vector<std::unique_ptr<Dev>> devs;
    for (int g = 0; g < ng; g++) devs.emplace_back(new Dev(g));
    std::vector<std::string> part(ng);
    on_devices(ng, [&](int g) {
      const size_t lo = n * (size_t)g / ng, hi = n * (size_t)(g + 1) / ng;
      if (lo == hi) return;
      chain2paf_range(*devs[g], recs.data() + lo, hi - lo, [&](Dev& dd, const uint8_t* d_out, size_t bytes) {
        const size_t at = part[g].size();
        part[g].resize(at + bytes);
        if (bytes) dd.download((uint8_t*)&part[g][at], d_out, bytes);
      });
    });
    for (const std::string& t : part) out.write(t);
  }
  out.close();
  return 0;
}

/* ---- maf2chain (converter.rs:57-91) ---------------------------------------------------------------------
 * GPU: K3 column-pair runs -> packed ops (wga_maf_runs_ops) -> data lines and trims (wga_cigar_chain; '=' and X
 * runs add up into one block like cigar_cat's M).  Host: chain headers (chain.rs:103-140,185-203). */
/* the chains of blocks recs[0 .. n), whose rows stand on device d (p); chain ids from chain_id0.  The text stays on the device:
 * *d_text, *bytes. */
static void maf2chain_text(Dev& d, const MafRows& p, const MafRecord* const* recs, uint32_t n, uint64_t chain_id0,
                           const uint8_t** d_text, uint64_t* bytes) {
  const uint8_t* d_rows = p.d_rows;
  auto *d_t = p.d_t, *d_q = p.d_q, *d_c = p.d_c;
  auto* d_s = p.d_s;
  auto* d_counts = (wga_cigar_counts*)d.alloc((size_t)n * sizeof(wga_cigar_counts));
  auto* d_cnt = (uint64_t*)d.alloc((size_t)n * 8);
  d.check(wga_maf_pair_stat(d.ctx, n, d_rows, d_t, d_q, d_c, d_s, d_counts, d_cnt, nullptr, nullptr));
  auto* d_roff = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
  d.check(wga_exclusive_scan_u64(d.ctx, n, d_cnt, d_roff));
  uint64_t n_runs = 0;
  d.download(&n_runs, d_roff + n, 1);
  auto* d_runs = (uint64_t*)d.alloc((n_runs + 1) * 8);
  d.check(wga_maf_pair_stat(d.ctx, n, d_rows, d_t, d_q, d_c, d_s, d_counts, d_cnt, d_runs, d_roff));
  auto* d_ocnt = (uint64_t*)d.alloc((size_t)n * 8);
  d.check(wga_maf_runs_ops(d.ctx, n, n_runs, d_runs, d_roff, d_c, d_ocnt, nullptr, nullptr));
  auto* d_ooff = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
  d.check(wga_exclusive_scan_u64(d.ctx, n, d_ocnt, d_ooff));
  uint64_t n_ops = 0;
  d.download(&n_ops, d_ooff + n, 1);
  auto* d_ops = (uint32_t*)d.alloc((n_ops + 4) * 4);
  d.check(wga_maf_runs_ops(d.ctx, n, n_runs, d_runs, d_roff, d_c, nullptr, d_ops, d_ooff));
  wga_cigar_batch cb;
  cb.d_ops = d_ops;
  cb.d_op_off = d_ooff;
  cb.d_strand_neg = d_s;
  cb.n_ops = n_ops;
  cb.n = n;
  auto* d_trim = (wga_chain_trim_t*)d.alloc((size_t)n * sizeof(wga_chain_trim_t));
  auto* d_nb = (uint64_t*)d.alloc((size_t)n * 8);
  auto* d_diag = (wga_rec_diag*)d.alloc((size_t)n * sizeof(wga_rec_diag));
  d.check(wga_cigar_chain(d.ctx, &cb, d_trim, d_nb, d_diag, nullptr, nullptr));
  std::vector<wga_chain_trim_t> trim(n);
  std::vector<uint64_t> nb(n);
  d.download(trim.data(), d_trim, n);
  d.download(nb.data(), d_nb, n);
  std::string blob;
  std::vector<uint64_t> blob_off{0}, dst, data_off(n);
  uint64_t pos = 0;
  for (uint32_t k = 0; k < n; k++) {
    const MafRecord& r = *recs[k];
    const wga_chain_trim_t& t = trim[k];
    const bool neg = r.q().neg;
    uint64_t qs = r.query_start(), qe = r.query_end();
    const uint64_t ts = r.t().start + t.head_del, te = r.t().start + r.t().align_size - t.tail_del;
    if (!neg) {
      qs += t.head_ins;
      qe -= t.tail_ins;
    } else { /* chain.rs:131-136: the new end is computed from the already updated start */
      qs = r.q().size - (qe - t.head_ins);
      qe = r.q().size - (qs + t.tail_ins);
    }
    std::string h = "chain\t255\t" + r.t().name + "\t";
    append_u64(h, r.t().size);
    h += "\t+\t";
    append_u64(h, ts);
    h.push_back('\t');
    append_u64(h, te);
    h += "\t" + r.q().name + "\t";
    append_u64(h, r.q().size);
    h += neg ? "\t-\t" : "\t+\t";
    append_u64(h, qs);
    h.push_back('\t');
    append_u64(h, qe);
    h.push_back('\t');
    append_u64(h, chain_id0 + (uint64_t)k);
    dst.push_back(pos);
    blob += h;
    blob_off.push_back(blob.size());
    pos += h.size();
    data_off[k] = pos;
    pos += nb[k];
    dst.push_back(pos);
    blob += "\n\n";
    blob_off.push_back(blob.size());
    pos += 2;
  }
  auto* d_out = (uint8_t*)d.alloc(pos + 64);
  d.check(wga_cigar_chain(d.ctx, &cb, nullptr, nullptr, nullptr, d_out, d.upload(data_off)));
  d.check(wga_scatter_bytes(d.ctx, 2 * n, d.upload((const uint8_t*)blob.data(), blob.size()), d.upload(blob_off), d_out,
                            d.upload(dst)));
  *d_text = d_out;
  *bytes = pos;
}

int cmd_maf2chain(const std::string* input, const std::string* query_name, Output& out) {
  Dev d;
  MafDevices md(d);
  MafChunks chunks(input);
  MafInput min;
  uint64_t chain_base = 0; /* chain id = index of the block in the whole input */
  std::string pending_error;
  while (pending_error.empty() && chunks.next(d, min)) {
  std::vector<MafRecord>& recs = min.recs;
  const uint64_t n_in_piece = recs.size();
  /* set_query_idx_byname fails per record, after the earlier records were written (:66-73) */
  size_t n_ok = recs.size();
  for (size_t k = 0; k < recs.size() && pending_error.empty(); k++) {
    MafRecord& r = recs[k];
    if (query_name) {
      size_t x = 0;
      for (; x < r.slines.size(); x++)
        if (r.slines[x].name == *query_name) break;
      if (x == r.slines.size()) {
        pending_error = "Query name:" + *query_name + " not found in MAF";
        n_ok = k;
        break;
      }
      r.query_idx = x;
    }
    if (r.query_idx >= r.slines.size()) {
      pending_error = "panic: MAF block with a single s-line has no query row (maf.rs:426 index out of bounds)";
      n_ok = k;
    }
  }
  recs.resize(n_ok);
  if (!recs.empty()) {
    const std::vector<const MafRecord*> all = all_records(recs);
    if (md.count() == 1) {
      md.run(min, all, false, [&](int, Dev& dg, const MafRows& p, uint32_t lo, uint32_t cnt) {
        const uint8_t* d_text = nullptr;
        uint64_t bytes = 0;
        maf2chain_text(dg, p, all.data() + lo, cnt, chain_base + lo, &d_text, &bytes);
        stream_out(dg, out, d_text, (size_t)bytes);
      });
    } else { /* --gpus N: a piece's blocks in contiguous ranges over the devices, the chains meet in block order */
      std::vector<std::string> part(md.count());
      md.run(min, all, false, [&](int g, Dev& dg, const MafRows& p, uint32_t lo, uint32_t cnt) {
        const uint8_t* d_text = nullptr;
        uint64_t bytes = 0;
        maf2chain_text(dg, p, all.data() + lo, cnt, chain_base + lo, &d_text, &bytes);
        part[g].resize((size_t)bytes);
        if (bytes) dg.download((uint8_t*)&part[g][0], d_text, bytes);
      });
      for (const std::string& t : part) out.write(t);
    }
  }
  chain_base += n_in_piece;
  md.release_all();
  }
  out.close();
  if (!pending_error.empty()) fail(pending_error);
  return 0;
}

/* ---- dotplot --out-format csv (tools/dotplot.rs) -----------------------------------
 * base-level: segments from wga_cigar_dotplot (PAF: device tokeniser; MAF: K3 runs -> wga_maf_runs_ops);
 * overview: one row per record, identity = matched / target_align_size from K1 / K3.  All data is generated
 * before anything is written (:208-262), so an error leaves the output empty.  The html / json outputs embed
 * the reference's Vega-Lite document and are not provided. */
/* the records of one device's share of a piece, as the csv rows need them */
struct DotRecs {
  std::vector<std::string> t_names, q_names;
  std::vector<uint64_t> ts, te, qs, qe, ali;
  std::vector<uint8_t> negs;
  void add(const std::string& t, const std::string& q, uint64_t ts_, uint64_t te_, uint64_t qs_, uint64_t qe_, bool neg, uint64_t a) {
    t_names.push_back(t);
    q_names.push_back(q);
    ts.push_back(ts_);
    te.push_back(te_);
    qs.push_back(qs_);
    qe.push_back(qe_);
    negs.push_back(neg ? 1 : 0);
    ali.push_back(a);
  }
};
/* csv rows (no header line) of the records R: base-level segments from wga_cigar_dotplot over the device batch cb, or one
 * overview row per record from the counts */
static std::string dotplot_rows(Dev& d, bool base, bool no_identity, uint64_t cutoff, const DotRecs& R, const wga_cigar_batch& cb,
                                const std::vector<wga_cigar_counts>& counts) {
  std::string text;
  const uint32_t n = (uint32_t)R.t_names.size();
  if (base) {
    if (n) {
      auto* d_cnt = (uint64_t*)d.alloc((size_t)n * 8);
      auto *d_ts = d.upload(R.ts), *d_qs = d.upload(R.qs);
      d.check(wga_cigar_dotplot(d.ctx, &cb, cutoff, d_ts, d_qs, d_cnt, nullptr, nullptr));
      auto* d_off = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
      d.check(wga_exclusive_scan_u64(d.ctx, n, d_cnt, d_off));
      std::vector<uint64_t> off(n + 1);
      d.download(off.data(), d_off, n + 1);
      auto* d_segs = (uint64_t*)d.alloc((off[n] + 1) * 5 * 8);
      d.check(wga_cigar_dotplot(d.ctx, &cb, cutoff, d_ts, d_qs, nullptr, d_segs, d_off));
      std::vector<uint64_t> segs(off[n] * 5);
      if (off[n]) d.download(segs.data(), d_segs, off[n] * 5);
      for (uint32_t k = 0; k < n; k++) {
        std::string names;
        names.push_back(',');
        append_csv_field(names, R.t_names[k], ',');
        names.push_back(',');
        append_csv_field(names, R.q_names[k], ',');
        names.push_back('\n');
        for (uint64_t x = off[k]; x < off[k + 1]; x++) {
          const uint64_t* sg = &segs[5 * x];
          for (int f = 0; f < 4; f++) {
            append_u64(text, sg[f]);
            text.push_back(',');
          }
          text.push_back("MID"[sg[4]]);
          text += names;
        }
      }
    }
  } else {
    for (uint32_t k = 0; k < n; k++) {
      const uint64_t a[] = {R.ts[k], R.te[k], R.negs[k] ? R.qe[k] : R.qs[k], R.negs[k] ? R.qs[k] : R.qe[k]}; /* dotplot.rs:400-406 */
      for (uint64_t v : a) {
        append_u64(text, v);
        text.push_back(',');
      }
      text += format_f64(no_identity ? 1.0 : (double)counts[k].match / (double)R.ali[k]);
      text.push_back(',');
      append_csv_field(text, R.t_names[k], ',');
      text.push_back(',');
      append_csv_field(text, R.q_names[k], ',');
      text.push_back('\n');
    }
  }
  return text;
}
/* records [lo, hi) of a PAF piece on device d; `err` = the reference's message for the first failing record of the range */
static std::string dotplot_paf_part(Dev& d, const PafInput& pin, size_t lo, size_t hi, const uint8_t* d_text_here, bool base,
                                    bool no_identity, uint64_t cutoff, std::string& err) {
  DotRecs R;
  for (size_t k = lo; k < hi; k++) {
    const PafRecord& r = pin.recs[k];
    R.add(r.target_name, r.query_name, r.target_start, r.target_end, r.query_start, r.query_end, r.neg, r.target_end - r.target_start);
  }
  wga_cigar_batch cb;
  cb.n = 0;
  std::vector<wga_cigar_counts> counts;
  const uint32_t n = (uint32_t)(hi - lo);
  if (n && (base || !no_identity)) {
    d.init();
    CigarTexts cigars;
    err = device_tokenise(d, pin, lo, n, cigars, &cb, nullptr, nullptr, d_text_here);
    if (!base && cb.n) { /* get_stat (paf.rs:205-209): ops outside M = X I D are an error */
      auto* d_counts = (wga_cigar_counts*)d.alloc((size_t)cb.n * sizeof(wga_cigar_counts));
      auto* d_diag = (wga_rec_diag*)d.alloc((size_t)cb.n * sizeof(wga_rec_diag));
      d.check(wga_cigar_stat(d.ctx, &cb, d_counts, d_diag, nullptr));
      counts.resize(cb.n);
      std::vector<wga_rec_diag> diag(cb.n);
      d.download(counts.data(), d_counts, cb.n);
      d.download(diag.data(), d_diag, cb.n);
      for (uint32_t k = 0; k < cb.n; k++)
        if (diag[k].bad_op_idx != WGA_NONE) {
          err = "CIGAR OP `" + cigar_op_token_at(cigars[k], diag[k].bad_op_idx) + "` invalid";
          break;
        }
    }
    if (!err.empty()) return std::string();
  }
  return dotplot_rows(d, base, no_identity, cutoff, R, cb, counts);
}
/* blocks recs[0 .. n) of a MAF piece, whose rows stand on device d (p) */
static std::string dotplot_maf_part(Dev& d, const MafRows& p, const MafRecord* const* recs, uint32_t n, bool base, bool no_identity,
                                    uint64_t cutoff) {
  DotRecs R;
  for (uint32_t k = 0; k < n; k++) {
    const MafRecord& r = *recs[k];
    R.add(r.t().name, r.q().name, r.t().start, r.t().start + r.t().align_size, r.query_start(), r.query_end(), r.q().neg,
          r.t().align_size);
  }
  wga_cigar_batch cb;
  cb.n = 0;
  std::vector<wga_cigar_counts> counts;
  if (n && (base || !no_identity)) {
    const uint8_t* d_rows = p.d_rows;
    auto *d_t = p.d_t, *d_q = p.d_q, *d_c = p.d_c;
    auto* d_s = p.d_s;
    auto* d_counts = (wga_cigar_counts*)d.alloc((size_t)n * sizeof(wga_cigar_counts));
    auto* d_cnt = (uint64_t*)d.alloc((size_t)n * 8);
    d.check(wga_maf_pair_stat(d.ctx, n, d_rows, d_t, d_q, d_c, d_s, d_counts, d_cnt, nullptr, nullptr));
    if (!base) {
      counts.resize(n);
      d.download(counts.data(), d_counts, n);
    } else {
      auto* d_roff = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
      d.check(wga_exclusive_scan_u64(d.ctx, n, d_cnt, d_roff));
      uint64_t n_runs = 0;
      d.download(&n_runs, d_roff + n, 1);
      auto* d_runs = (uint64_t*)d.alloc((n_runs + 1) * 8);
      d.check(wga_maf_pair_stat(d.ctx, n, d_rows, d_t, d_q, d_c, d_s, d_counts, d_cnt, d_runs, d_roff));
      auto* d_ocnt = (uint64_t*)d.alloc((size_t)n * 8);
      d.check(wga_maf_runs_ops(d.ctx, n, n_runs, d_runs, d_roff, d_c, d_ocnt, nullptr, nullptr));
      auto* d_ooff = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
      d.check(wga_exclusive_scan_u64(d.ctx, n, d_ocnt, d_ooff));
      uint64_t n_ops = 0;
      d.download(&n_ops, d_ooff + n, 1);
      auto* d_ops = (uint32_t*)d.alloc((n_ops + 4) * 4);
      d.check(wga_maf_runs_ops(d.ctx, n, n_runs, d_runs, d_roff, d_c, nullptr, d_ops, d_ooff));
      cb.d_ops = d_ops;
      cb.d_op_off = d_ooff;
      cb.d_strand_neg = d_s;
      cb.n_ops = n_ops;
      cb.n = n;
    }
  }
  return dotplot_rows(d, base, no_identity, cutoff, R, cb, counts);
}

int cmd_dotplot(const std::string* input, const std::string& format, const std::string& out_format,
                const std::string& mode, bool no_identity, uint64_t cutoff, const std::string* query_name, Output& out) {
  if (mode != "base-level" && mode != "overview") fail("invalid value '" + mode + "' for '--mode <MODE>'");
  if (out_format != "csv") {
    if (out_format == "html" || out_format == "json")
      fail("out-format `" + out_format + "` embeds the reference's Vega-Lite document and is not provided by this engine (use --out-format csv)");
    fail("invalid value '" + out_format + "' for '--out-format <OUT_FORMAT>'");
  }
  if (format != "maf" && format != "paf") fail("Only support MAF and PAF format");
  const bool base = mode == "base-level";
  std::string text; /* all data is generated before anything is written (dotplot.rs:208-262) */
  uint64_t n_records = 0;
  Dev d;
  /* --gpus N: a piece's records / blocks in contiguous ranges over the devices; the rows meet in input order */
  if (format == "paf") {
    PafChunks chunks(input, false);
    PafInput pin;
    std::vector<std::unique_ptr<Dev>> devs;
    while (chunks.next(d, pin)) {
      const size_t n = pin.recs.size();
      n_records += n;
      const int ng = g_gpus;
      std::vector<std::string> part(ng), err(ng);
      if (ng == 1) {
        part[0] = dotplot_paf_part(d, pin, 0, n, nullptr, base, no_identity, cutoff, err[0]);
      } else {
        d.init();
        if (devs.empty()) {
          devs.emplace_back(new Dev(0));
          devs[0]->ctx = d.ctx; /* device 0's context is the reader's */
          devs[0]->own_ctx = false;
          for (int g = 1; g < ng; g++) devs.emplace_back(new Dev(g));
        }
        std::string text16;
        if (pin.on_device) text16 = pin.text + std::string(16, '\0');
        on_devices(ng, [&](int g) {
          const size_t lo = n * (size_t)g / ng, hi = n * (size_t)(g + 1) / ng;
          if (lo == hi) return;
          Dev& dg = *devs[g];
          dg.init();
          const size_t keep = dg.owned.size();
          const uint8_t* d_text = nullptr;
          if (pin.on_device) d_text = g == 0 ? pin.d_text : dg.upload((const uint8_t*)text16.data(), text16.size());
          part[g] = dotplot_paf_part(dg, pin, lo, hi, d_text, base, no_identity, cutoff, err[g]);
          dg.release_to(keep);
        });
      }
      for (int g = 0; g < ng; g++) {
        if (!err[g].empty()) {
          out.close();
          fail(err[g]);
        }
        text += part[g];
      }
      d.release_all();
    }
  } else {
    MafDevices md(d);
    MafChunks chunks(input);
    MafInput min;
    while (chunks.next(d, min)) {
      std::vector<MafRecord>& recs = min.recs;
      select_query(recs, query_name);
      n_records += recs.size();
      if (!recs.empty()) {
        const std::vector<const MafRecord*> all = all_records(recs);
        std::vector<std::string> part(md.count());
        if (base || !no_identity) {
          md.run(min, all, false, [&](int g, Dev& dg, const MafRows& p, uint32_t lo, uint32_t cnt) {
            part[g] = dotplot_maf_part(dg, p, all.data() + lo, cnt, base, no_identity, cutoff);
          });
        } else {
          part[0] = dotplot_maf_part(d, MafRows(), all.data(), (uint32_t)all.size(), base, no_identity, cutoff);
        }
        for (const std::string& t : part) text += t;
      }
      md.release_all();
    }
  }
  /* the header line: base-level once a segment exists, overview once a record does */
  if (base ? !text.empty() : n_records != 0)
    text = std::string(base ? "ref_start,ref_end,query_start,query_end,cigar,ref_chro,query_chro\n"
                            : "ref_start,ref_end,query_start,query_end,identity,ref_chro,query_chro\n") + text;
  out.write(text);
  out.close();
  return 0;
}

/* ---- pafcov (pafcov.rs:13-83) --------------------------------------------------------------------- */
/* `wgatools --gpus N pafcov`: a target's coverage array lives on device fnv1a64(target_name) % N, which accumulates the
 * records of its targets, turns the marks into counts and formats its targets' BED text (pafcov.rs:18-64).  The text is
 * written in the reference's target order: a first pass asks every device for the byte count of each of its 4 M-position
 * chunks, the prefix over (target, chunk) gives every chunk its file offset, the second pass formats and pwrite()s.  No
 * collective: the element-wise merge of the reference's per-thread arrays (pafcov.rs:29-53) has nothing left to merge. */
int cmd_pafcov_multi(const std::string* input, Output& out, int ngpu, bool spread) {
  std::vector<std::unique_ptr<Dev>> devs;
  for (int g = 0; g < ngpu; g++) devs.emplace_back(new Dev(g));
  Dev& d0 = *devs[0];
  std::vector<std::string> targets;
  std::unordered_map<std::string, uint32_t> tid;
  std::vector<uint64_t> cov_len;
  uint64_t n_records = 0;
  auto note_targets = [&](const std::vector<PafRecord>& recs) {
    for (const auto& r : recs) {
      if (tid.find(r.target_name) == tid.end()) {
        tid.emplace(r.target_name, (uint32_t)targets.size());
        targets.push_back(r.target_name);
        cov_len.push_back(r.target_length);
      }
    }
    n_records += recs.size();
  };
  PafInput whole;
  bool single = false;
  if (!input) {
    whole = load_paf(d0, input, false);
    note_targets(whole.recs);
    single = true;
  } else {
    PafChunks first(input, false);
    PafInput pin;
    if (first.next(d0, whole)) {
      note_targets(whole.recs);
      if (!first.next(d0, pin)) {
        single = true;
      } else {
        if (whole.d_text) d0.release(whole.d_text);
        whole = PafInput();
        do {
          note_targets(pin.recs);
          d0.release_all();
        } while (first.next(d0, pin));
      }
    }
  }
  const uint32_t nt = (uint32_t)targets.size();
  uint64_t pos0 = 0;
  const int fd = out.plain_fd(&pos0);
  if (fd < 0) fail("internal error: --gpus needs a plain output file");
  uint64_t file_end = pos0;
  if (n_records) {
    /* the targets of every device, in the reference's order, and its coverage arrays */
    std::vector<int> owner(nt);
    std::vector<uint32_t> local(nt);
    std::vector<std::vector<uint64_t>> off_g(ngpu), len_g(ngpu);
    std::vector<uint64_t> total_g(ngpu, 0);
    for (uint32_t t = 0; t < nt; t++) {
      const int g = (int)(fnv1a64(targets[t]) % (uint64_t)ngpu);
      owner[t] = g;
      for (int h = 0; h < ngpu; h++) {
        if (!spread && h != g) continue; /* --spread: every device holds (its share of) every target */
        if (h == g || spread) local[t] = (uint32_t)off_g[h].size();
        off_g[h].push_back(total_g[h]);
        len_g[h].push_back(cov_len[t]);
        total_g[h] += (cov_len[t] + 3) & ~3ull;
      }
    }
    std::vector<int32_t*> d_cov(ngpu, nullptr);
    std::vector<uint64_t*> d_off(ngpu, nullptr), d_len(ngpu, nullptr);
    std::vector<size_t> keep(ngpu, 0);
    on_devices(ngpu, [&](int g) {
      Dev& d = *devs[g];
      d.init();
      if (off_g[g].empty()) return;
      d_cov[g] = (int32_t*)d.alloc((total_g[g] + 4) * 4);
      d.check(wga_memset(d.ctx, d_cov[g], 0, (total_g[g] + 4) * 4));
      d_off[g] = d.upload(off_g[g]);
      d_len[g] = d.upload(len_g[g]);
    });
    for (int g = 0; g < ngpu; g++) keep[g] = devs[g]->owned.size();
    uint64_t recs_seen = 0;
    auto accumulate = [&](const PafInput& pin) {
      const size_t n = pin.recs.size();
      std::vector<std::vector<size_t>> mine(ngpu);
      for (size_t i = 0; i < n; i++) /* --spread deals the records out round robin (one hot target: hash sharding would not spread it) */
        mine[spread ? (int)((recs_seen + i) % (uint64_t)ngpu) : owner[tid[pin.recs[i].target_name]]].push_back(i);
      recs_seen += n;
      std::vector<size_t> bad_at(ngpu, n);
      std::vector<std::string> bad_msg(ngpu);
      std::string text16;
      if (pin.on_device && ngpu > 1) text16 = pin.text + std::string(16, '\0');
      on_devices(ngpu, [&](int g) {
        Dev& d = *devs[g];
        const std::vector<size_t>& w = mine[g];
        if (w.empty()) return;
        const size_t mark = d.owned.size();
        const uint8_t* d_text = nullptr;
        if (pin.on_device) d_text = g == 0 ? pin.d_text : d.upload((const uint8_t*)text16.data(), text16.size());
        std::vector<uint64_t> t_start;
        std::vector<uint32_t> target_id;
        for (size_t i : w) {
          target_id.push_back(local[tid[pin.recs[i].target_name]]);
          t_start.push_back(pin.recs[i].target_start);
        }
        CigarTexts cigars;
        wga_cigar_batch cb;
        const std::string terr = device_tokenise(d, pin, 0, (uint32_t)w.size(), cigars, &cb, nullptr, w.data(), d_text);
        if (!terr.empty()) { /* update_cov_vec takes every op char: only the tokeniser can fail */
          bad_at[g] = w[cb.n];
          bad_msg[g] = terr;
          return;
        }
        d.check(wga_pafcov_accumulate(d.ctx, &cb, d.upload(target_id), d.upload(t_start), d_off[g], d_len[g], d_cov[g], total_g[g]));
        d.check(wga_sync(d.ctx));
        if (g != 0) d.release_to(mark); /* device 0: the reader releases its piece */
      });
      size_t first_bad = n;
      for (int g = 0; g < ngpu; g++) first_bad = std::min(first_bad, bad_at[g]);
      if (first_bad < n)
        for (int g = 0; g < ngpu; g++)
          if (bad_at[g] == first_bad) fail(bad_msg[g]); /* buffered driver: nothing is written */
    };
    if (single) {
      accumulate(whole);
    } else {
      PafChunks second(input, false);
      PafInput pin;
      while (second.next(d0, pin)) {
        accumulate(pin);
        d0.check(wga_sync(d0.ctx));
        d0.release_to(keep[0]);
      }
    }
    on_devices(ngpu, [&](int g) {
      Dev& d = *devs[g];
      if (off_g[g].empty()) return;
      d.check(wga_pafcov_finalize(d.ctx, (uint32_t)off_g[g].size(), d_off[g], d_len[g], d_cov[g]));
    });
    /* --spread: every device holds partial counts of every target (the scan is linear: partial marks -> partial counts);
     * one reduce-scatter over the whole counter space leaves device g with the summed slice g (wga_reduce_scatter_i32) */
    std::vector<uint64_t> slice_lo(ngpu + 1, 0);
    if (spread) {
      std::vector<wga_ctx*> cx(ngpu);
      std::vector<int32_t*> bufs(ngpu);
      for (int g = 0; g < ngpu; g++) {
        cx[g] = devs[g]->ctx;
        bufs[g] = d_cov[g];
        slice_lo[g] = total_g[0] * (uint64_t)g / (uint64_t)ngpu;
      }
      slice_lo[ngpu] = total_g[0];
      if (wga_reduce_scatter_i32(cx.data(), ngpu, bufs.data(), total_g[0])) fail(std::string("GPU engine: ") + wga_last_error());
    }
    /* BED text, a few million positions at a time (pafcov.rs:56-60): sizes, offsets, then text at its place.  A chunk =
     * positions [pos, pos + cnt) of target t on the device that holds their counts. */
    const uint32_t kChunk = 4u << 20;
    struct Chunk {
      uint32_t t, cnt;
      uint64_t pos, bytes, off;
      int g;
    };
    std::vector<Chunk> chunks_all;
    for (uint32_t t = 0; t < nt; t++) {
      uint64_t pos = 0;
      while (pos < cov_len[t]) {
        uint64_t cnt = std::min<uint64_t>(kChunk, cov_len[t] - pos);
        int g = owner[t];
        if (spread) { /* cut at the slice boundary of the counter space */
          const uint64_t x = off_g[0][t] + pos;
          g = 0;
          while (g + 1 < ngpu && slice_lo[g + 1] <= x) g++;
          cnt = std::min<uint64_t>(cnt, slice_lo[g + 1] - x);
        }
        chunks_all.push_back(Chunk{t, (uint32_t)cnt, pos, 0, 0, g});
        pos += cnt;
      }
    }
    auto format_pass = [&](bool fill) {
      on_devices(ngpu, [&](int g) {
        Dev& d = *devs[g];
        if (off_g[g].empty()) return;
        const size_t mark = d.owned.size();
        auto* d_loff = (uint64_t*)d.alloc(((size_t)kChunk + 1) * 8);
        uint8_t* d_txt = nullptr;
        uint64_t txt_cap = 0;
        uint32_t name_of = 0xFFFFFFFFu;
        uint8_t* d_name = nullptr;
        for (Chunk& c : chunks_all) {
          if (c.g != g) continue;
          if (name_of != c.t) {
            d_name = d.upload((const uint8_t*)targets[c.t].data(), targets[c.t].size());
            name_of = c.t;
          }
          const int32_t* cp = d_cov[g] + off_g[g][local[c.t]] + c.pos;
          d.check(wga_pafcov_format(d.ctx, d_name, (uint32_t)targets[c.t].size(), cp, c.pos, c.cnt, d_loff, nullptr));
          uint64_t bytes = 0;
          d.download(&bytes, d_loff + c.cnt, 1);
          if (!fill) {
            c.bytes = bytes;
            continue;
          }
          if (bytes > txt_cap) {
            if (d_txt) d.release(d_txt);
            txt_cap = bytes + bytes / 4;
            d_txt = (uint8_t*)d.alloc(txt_cap);
          }
          d.check(wga_pafcov_format(d.ctx, d_name, (uint32_t)targets[c.t].size(), cp, c.pos, c.cnt, d_loff, d_txt));
          d.check(wga_sync(d.ctx));
          write_runs(d, fd, d_txt, {{0, bytes, c.off}});
        }
        d.release_to(mark);
      });
    };
    format_pass(false);
    for (Chunk& c : chunks_all) {
      c.off = file_end;
      file_end += c.bytes;
    }
    format_pass(true);
  }
  out.advance(file_end - pos0);
  out.close();
  return 0;
}

int cmd_pafcov(const std::string* input, Output& out, bool spread) {
  {
    uint64_t pos = 0;
    if (g_gpus > 1 && out.plain_fd(&pos) >= 0) return cmd_pafcov_multi(input, out, g_gpus, spread);
  }
  Dev d;
  /* A file is read twice in line-aligned pieces — first for the targets (names in first-appearance order, array
   * length = target_length of the first record seen), then to accumulate — so that only one piece of text is held at
   * a time; stdin cannot be read twice and is taken whole. */
  std::vector<std::string> targets; /* first-appearance order (the reference: HashMap order) */
  std::unordered_map<std::string, uint32_t> tid;
  std::vector<uint64_t> cov_len;
  uint64_t n_records = 0;
  auto note_targets = [&](const std::vector<PafRecord>& recs) {
    for (const auto& r : recs) {
      if (tid.find(r.target_name) == tid.end()) {
        tid.emplace(r.target_name, (uint32_t)targets.size());
        targets.push_back(r.target_name);
        cov_len.push_back(r.target_length);
      }
    }
    n_records += recs.size();
  };
  PafInput whole;
  bool single = false; /* the input is one piece: kept, not read again */
  if (!input) {
    whole = load_paf(d, input, false);
    note_targets(whole.recs);
    single = true;
  } else {
    PafChunks first(input, false);
    PafInput pin;
    if (first.next(d, whole)) {
      note_targets(whole.recs);
      if (!first.next(d, pin)) {
        single = true;
      } else {
        if (whole.d_text) d.release(whole.d_text);
        whole = PafInput();
        do {
          note_targets(pin.recs);
          d.release_all();
        } while (first.next(d, pin));
      }
    }
  }
  const uint32_t nt = (uint32_t)targets.size();
  if (n_records) {
    std::vector<uint64_t> cov_off(nt);
    uint64_t total = 0;
    for (uint32_t t = 0; t < nt; t++) {
      cov_off[t] = total;
      total += (cov_len[t] + 3) & ~3ull;
    }
    d.init();
    auto* d_cov = (int32_t*)d.alloc((total + 4) * 4);
    d.check(wga_memset(d.ctx, d_cov, 0, (total + 4) * 4));
    auto *d_off = d.upload(cov_off), *d_len = d.upload(cov_len);
    const size_t keep = d.owned.size();
    auto accumulate = [&](const PafInput& pin) {
      const std::vector<PafRecord>& recs = pin.recs;
      const uint32_t n = (uint32_t)recs.size();
      std::vector<uint64_t> t_start;
      std::vector<uint32_t> target_id;
      for (const auto& r : recs) {
        target_id.push_back(tid[r.target_name]);
        t_start.push_back(r.target_start);
      }
      CigarTexts cigars;
      wga_cigar_batch cb;
      const std::string terr = device_tokenise(d, pin, 0, n, cigars, &cb); /* update_cov_vec takes every op char */
      if (!terr.empty()) fail(terr);
      d.check(wga_pafcov_accumulate(d.ctx, &cb, d.upload(target_id), d.upload(t_start), d_off, d_len, d_cov, total));
    };
    if (single) {
      accumulate(whole);
    } else {
      PafChunks second(input, false);
      PafInput pin;
      while (second.next(d, pin)) {
        accumulate(pin);
        d.check(wga_sync(d.ctx));
        while (d.owned.size() > keep) d.release(d.owned.back()); /* this piece's text and buffers */
      }
    }
    d.check(wga_pafcov_finalize(d.ctx, nt, d_off, d_len, d_cov));
    /* the BED text is formatted on the device, a few million positions at a time (pafcov.rs:56-60) */
    const uint32_t kChunk = 4u << 20;
    auto* d_loff = (uint64_t*)d.alloc(((size_t)kChunk + 1) * 8);
    uint8_t* d_txt = nullptr;
    uint64_t txt_cap = 0;
    std::string text;
    for (uint32_t t = 0; t < nt; t++) {
      auto* d_name = d.upload((const uint8_t*)targets[t].data(), targets[t].size());
      for (uint64_t pos = 0; pos < cov_len[t]; pos += kChunk) {
        const uint32_t cnt = (uint32_t)std::min<uint64_t>(kChunk, cov_len[t] - pos);
        const int32_t* cp = d_cov + cov_off[t] + pos;
        d.check(wga_pafcov_format(d.ctx, d_name, (uint32_t)targets[t].size(), cp, pos, cnt, d_loff, nullptr));
        uint64_t bytes = 0;
        d.download(&bytes, d_loff + cnt, 1);
        if (bytes > txt_cap) {
          if (d_txt) d.release(d_txt);
          txt_cap = bytes + bytes / 4;
          d_txt = (uint8_t*)d.alloc(txt_cap);
        }
        d.check(wga_pafcov_format(d.ctx, d_name, (uint32_t)targets[t].size(), cp, pos, cnt, d_loff, d_txt));
        stream_out(d, out, d_txt, (size_t)bytes); /* 4 M positions of BED text per piece, copied and written in overlap */
      }
    }
    text.clear();
    out.write(text);
  }
  out.close();
  return 0;
}


/* ---- pafpseudo (pseudomaf.rs:18-237) ----------------------------------------------------------------
 * Host: grouping by target / query, insertion sort by target_start, gap / overlap / contained
 * logic (:86-95,:147-175,:198-202).  GPU: every kept record's segment in target coordinates
 * (gen_pesudo_maf_by_cigar + the head trim), all targets in one batch. */
struct PseudoSeg {
  size_t rec;        /* index into recs */
  uint64_t gap;      /* '-' columns written before the segment */
  uint64_t overlap;  /* leading columns dropped */
};
struct PseudoQuery {
  std::string name;
  uint64_t size = 0;
  std::vector<size_t> recs; /* sorted by target_start */
  std::vector<PseudoSeg> segs;
  uint64_t tail = 0;
};
struct PseudoTarget {
  std::string name;
  std::vector<size_t> recs;
  std::vector<PseudoQuery> queries;
  uint64_t target_size_first = 0;
};

/* slice::binary_search_by of the Rust std the reference was built with: probe the middle, return
 * at the first equal element [toolchain-dependent for ties: unpinned] */
size_t rust_binary_search_pos(const std::vector<size_t>& v, const std::vector<PafRecord>& recs, uint64_t key) {
  size_t size = v.size(), left = 0, right = size;
  while (left < right) {
    size_t mid = left + size / 2;
    uint64_t probe = recs[v[mid]].target_start;
    if (probe == key) return mid;
    if (probe < key)
      left = mid + 1;
    else
      right = mid;
    size = right - left;
  }
  return left;
}

int cmd_pafpseudo(const std::string* input, const std::string& outdir, bool rewrite, const std::string* fasta,
                  const std::string* only_target) {
  if (outdir == "-") fail("Stdout not allowed here"); /* errors.rs:37 */
  struct stat st;
  if (stat(outdir.c_str(), &st) != 0) {
    /* create_dir_all: every missing component, no shell involved */
    for (size_t k = 1; k <= outdir.size(); k++)
      if (k == outdir.size() || outdir[k] == '/') {
        const std::string part = outdir.substr(0, k);
        if (mkdir(part.c_str(), 0777) != 0 && errno != EEXIST)
          fail("IO error:cannot create directory `" + outdir + "`: " + strerror(errno));
      }
  } else {
    if (!S_ISDIR(st.st_mode)) fail("Path `" + outdir + "` is not a dir");
    if (!rewrite) fail("File `" + outdir + "` already exists, please add `-r` to rewrite it.");
  }
  /* --gpus N: a target's records live on device fnv1a64(target_name) % N (the reference already works target by target,
   * pseudomaf.rs:62-72); every device tokenises, sums and fills its records, the walk and every check run on the host in the
   * reference's processing order, so the first error is the same whatever N is */
  const int ngpu = g_gpus > 1 ? g_gpus : 1;
  std::vector<std::unique_ptr<Dev>> devs;
  for (int g = 0; g < ngpu; g++) devs.emplace_back(new Dev(g));
  Dev& d = *devs[0];
  PafInput pin = load_paf(d, input, false);
  const std::vector<PafRecord>& recs = pin.recs;
  const bool base = fasta != nullptr;
  std::vector<DevFasta> fas(ngpu);
  DevFasta& fa = fas[0]; /* the index (names -> pool offsets) is the same on every device */
  const uint32_t n_all = (uint32_t)recs.size();
  std::vector<uint8_t> has_tag(n_all, 1);
  /* records without a tag are given an empty span so that a device's batch keeps file order */
  if (pin.on_device) {
    for (uint32_t i = 0; i < n_all; i++)
      if (pin.cg_beg[i] == WGA_NONE) {
        has_tag[i] = 0;
        pin.cg_beg[i] = pin.cg_end[i] = 0;
      }
  } else {
    for (uint32_t i = 0; i < n_all; i++) {
      int err = 0;
      (void)paf_cigar_string(recs[i], &err);
      if (err) {
        has_tag[i] = 0;
        pin.recs[i].tags.push_back("cg:Z:"); /* placeholder: tokenises to the empty-CIGAR error, never used */
      }
    }
  }
  std::vector<std::vector<size_t>> mine(ngpu);
  std::vector<uint32_t> owner(n_all, 0), at(n_all, 0); /* record i = record at[i] of device owner[i] */
  for (uint32_t i = 0; i < n_all; i++) {
    const int g = ngpu > 1 ? (int)(fnv1a64(recs[i].target_name) % (uint64_t)ngpu) : 0;
    owner[i] = (uint32_t)g;
    at[i] = (uint32_t)mine[g].size();
    mine[g].push_back(i);
  }
  /* every device: pools, its records' CIGARs tokenised in file order; a record's tag / tokeniser error only counts when
   * the walk below reaches that record (the reference parses a CIGAR when it processes the record) */
  std::vector<CigarTexts> cigars(ngpu);
  std::vector<wga_cigar_batch> cbs(ngpu);
  std::vector<std::vector<wga_tok_err>> terrs(ngpu);
  std::string text16;
  if (pin.on_device && ngpu > 1) text16 = pin.text + std::string(16, '\0');
  on_devices(ngpu, [&](int g) {
    Dev& dg = *devs[g];
    cbs[g].n = 0;
    if (base) {
      dg.init();
      fas[g].load(dg, *fasta);
    }
    if (mine[g].empty()) return;
    dg.init();
    const uint8_t* d_text = nullptr;
    if (pin.on_device) d_text = g == 0 ? pin.d_text : dg.upload((const uint8_t*)text16.data(), text16.size());
    (void)device_tokenise(dg, pin, 0, (uint32_t)mine[g].size(), cigars[g], &cbs[g], &terrs[g], ngpu > 1 ? mine[g].data() : nullptr,
                          d_text);
  });
  /* 1. group by target (:25-42), then by query with sorted insertion (:86-95) */
  std::vector<PseudoTarget> targets;
  std::unordered_map<std::string, size_t> tindex;
  for (size_t i = 0; i < recs.size(); i++) {
    const PafRecord& r = recs[i];
    if (only_target && r.target_name != *only_target) continue;
    auto it = tindex.find(r.target_name);
    if (it == tindex.end()) {
      it = tindex.emplace(r.target_name, targets.size()).first;
      targets.emplace_back();
      targets.back().name = r.target_name;
    }
    targets[it->second].recs.push_back(i);
  }
  std::vector<uint64_t> q_off(n_all, 0), q_len(n_all, 0), skip(n_all, UINT64_MAX); /* skip = all: a record the walk drops */
  for (auto& t : targets) {
    std::unordered_map<std::string, size_t> qindex;
    for (size_t i : t.recs) {
      const PafRecord& r = recs[i];
      auto it = qindex.find(r.query_name);
      if (it == qindex.end()) {
        it = qindex.emplace(r.query_name, t.queries.size()).first;
        t.queries.emplace_back();
        t.queries.back().name = r.query_name;
      }
      PseudoQuery& q = t.queries[it->second];
      size_t pos = rust_binary_search_pos(q.recs, recs, r.target_start);
      q.recs.insert(q.recs.begin() + (long)pos, i);
    }
    /* 2. walk (:108-205) */
    uint64_t target_size = 0;
    bool first = true;
    for (auto& q : t.queries) {
      uint64_t last_target_end = 0;
      bool first_query = true;
      for (size_t i : q.recs) {
        const PafRecord& r = recs[i];
        target_size = r.target_length;
        if (first) {
          t.target_size_first = target_size;
          first = false;
        }
        if (first_query) q.size = r.query_length;
        uint64_t overlap = 0, gap = 0;
        if (r.target_start > last_target_end) {
          gap = r.target_start - last_target_end;
        } else {
          if (last_target_end > r.target_end) continue; /* contained: dropped */
          overlap = last_target_end - r.target_start;
        }
        last_target_end = r.target_end;
        uint64_t qo = 0, ql = 0;
        if (base) fa.fetch(q.name, r.query_start, r.query_end - 1, &qo, &ql); /* :222-225 */
        if (!has_tag[i]) fail("CIGAR start tag not found"); /* errors.rs:57 */
        const wga_tok_err& te = terrs[owner[i]][at[i]];
        if (te.err) fail(cigar_error_message(te.err, cigars[owner[i]][at[i]], (size_t)te.tok_off, te.tok_len));
        q_off[i] = qo;
        q_len[i] = ql;
        skip[i] = overlap;
        q.segs.push_back(PseudoSeg{i, gap, overlap});
        first_query = false;
      }
      if (last_target_end > target_size)
        fail("panic: attempt to fill a negative tail (pseudomaf.rs:198 underflows)");
      q.tail = target_size - last_target_end;
    }
  }
  /* 3. GPU: class sums -> segment lengths -> fill; a device's batch is its records in file order, dropped records write nothing */
  std::vector<uint64_t> seg_len(n_all, 0), dst_off(n_all, 0); /* dst_off: inside the owner's segment text */
  std::vector<std::string> segs(ngpu);
  if (n_all) {
    std::vector<wga_class_sums> sums(n_all);
    std::vector<wga_class_sums*> d_sums(ngpu, nullptr);
    on_devices(ngpu, [&](int g) {
      const uint32_t n = cbs[g].n;
      if (!n) return;
      Dev& dg = *devs[g];
      d_sums[g] = (wga_class_sums*)dg.alloc((size_t)n * sizeof(wga_class_sums));
      dg.check(wga_cigar_class_sums(dg.ctx, &cbs[g], d_sums[g]));
      std::vector<wga_class_sums> part(n);
      dg.download(part.data(), d_sums[g], n);
      for (uint32_t k = 0; k < n; k++) sums[mine[g][k]] = part[k];
    });
    /* in the order the reference processes the records (an error of an earlier one wins) */
    for (const auto& t : targets)
      for (const auto& q : t.queries)
        for (const auto& sg : q.segs) {
          const size_t k = sg.rec;
          uint64_t len = base ? q_len[k] - (sums[k].i + sums[k].s) + sums[k].d : sums[k].mx + sums[k].d;
          if (base && q_len[k] < sums[k].i + sums[k].s) len = 0; /* reported as a panic below */
          if (skip[k] > len) fail("panic: String::drain range out of bounds (pseudomaf.rs:191)");
          seg_len[k] = len - skip[k];
        }
    std::vector<std::vector<wga_rec_diag>> diag(ngpu);
    on_devices(ngpu, [&](int g) {
      const uint32_t n = cbs[g].n;
      if (!n) return;
      Dev& dg = *devs[g];
      std::vector<uint64_t> qo(n), ql(n), sk(n), doff(n + 1, 0);
      for (uint32_t k = 0; k < n; k++) {
        const size_t i = mine[g][k];
        qo[k] = q_off[i], ql[k] = q_len[i], sk[k] = skip[i];
        dst_off[i] = doff[k];
        doff[k + 1] = doff[k] + seg_len[i];
      }
      auto* d_out = (uint8_t*)dg.alloc(doff[n] + 64);
      auto* d_diag = (wga_rec_diag*)dg.alloc((size_t)n * sizeof(wga_rec_diag));
      uint8_t* d_pool = base ? fas[g].d_pool : nullptr;
      dg.check(wga_pafpseudo_fill(dg.ctx, &cbs[g], base ? 1 : 0, d_pool, fas[g].bytes, base ? dg.upload(qo) : nullptr,
                                  base ? dg.upload(ql) : nullptr, dg.upload(sk), d_out, dg.upload(doff), d_diag));
      diag[g].resize(n);
      dg.download(diag[g].data(), d_diag, n);
      segs[g].resize(doff[n]);
      if (doff[n]) dg.download((uint8_t*)segs[g].data(), d_out, doff[n]);
    });
    for (const auto& t : targets)
      for (const auto& q : t.queries)
        for (const auto& sg : q.segs) {
          const size_t k = sg.rec;
          const wga_rec_diag& dk = diag[owner[k]][at[k]];
          if (dk.bad_base_pos != WGA_NONE)
            fail(std::string("Invalid Base: `") + fas[owner[k]].at(*devs[owner[k]], q_off[k] + q_len[k] - 1 - dk.bad_base_pos) + "`");
          if (dk.panic_op_idx != WGA_NONE)
            fail("panic: String::drain / insert_str beyond the end of the query sequence (cigar.rs:772,779)");
        }
  }
  /* 4. one file per target (:62-72, :98-209) */
  for (const auto& t : targets) {
    Output out;
    out.open(outdir + "/" + t.name + ".maf", true);
    std::string text = "a score=0\n";
    if (!t.queries.empty()) {
      text += "s\t" + t.name + "\t0\t";
      append_u64(text, t.target_size_first);
      text += "\t+\t";
      append_u64(text, t.target_size_first);
      text.push_back('\t');
      if (base) {
        uint64_t o, l;
        fa.fetch(t.name, 0, t.target_size_first - 1, &o, &l);
        text += fa.slice(d, o, l);
      } else {
        text.append(t.target_size_first, 'N');
      }
      text.push_back('\n');
    }
    for (const auto& q : t.queries) {
      text += "s\t" + q.name + "\t0\t";
      append_u64(text, q.size);
      text += "\t+\t";
      append_u64(text, q.size);
      text.push_back('\t');
      for (const auto& sg : q.segs) {
        text.append(sg.gap, '-');
        text.append(segs[owner[sg.rec]], dst_off[sg.rec], seg_len[sg.rec]);
      }
      text.append(q.tail, '-');
      text.push_back('\n');
      if (text.size() > (1u << 26)) {
        out.write(text);
        text.clear();
      }
    }
    text.push_back('\n');
    out.write(text);
    out.close();
  }
  return 0;
}


/* ---- call (MAF) (caller.rs:42-265, 388-608) ------------------------------------------------------------
 * GPU: the column walk — runs of equal caller class with non-gap prefix counts (wga_maf_call_runs).
 * Host: everything that works on runs instead of columns — SV-safe chunk cuts, chunk coordinates,
 * the after_m event rules, REF/ALT slices and the VCF text (noodles-vcf 0.43 layout, README.md:323-343). */
struct CallRun {
  uint64_t start, tb, qb;
  uint32_t cls; /* 0 '=', 1 I, 2 D, 3 X, 4 W */
};
struct CallBlock {
  const MafRecord* rec;
  std::vector<CallRun> runs;
  uint64_t total;
  uint64_t end(size_t k) const { return k + 1 < runs.size() ? runs[k + 1].start : total; }
  size_t run_at(uint64_t col) const { /* run containing column col */
    size_t lo = 0, hi = runs.size();
    while (hi - lo > 1) {
      size_t mid = (lo + hi) / 2;
      if (runs[mid].start <= col)
        lo = mid;
      else
        hi = mid;
    }
    return lo;
  }
  static bool adv_t(uint32_t c) { return c == 0 || c == 3 || c == 2; }
  static bool adv_q(uint32_t c) { return c == 0 || c == 3 || c == 1; }
  uint64_t t_before(uint64_t col) const {
    if (col >= total) return runs.empty() ? 0 : runs.back().tb + (adv_t(runs.back().cls) ? total - runs.back().start : 0);
    size_t k = run_at(col);
    return runs[k].tb + (adv_t(runs[k].cls) ? col - runs[k].start : 0);
  }
  uint64_t q_before(uint64_t col) const {
    if (col >= total) return runs.empty() ? 0 : runs.back().qb + (adv_q(runs.back().cls) ? total - runs.back().start : 0);
    size_t k = run_at(col);
    return runs[k].qb + (adv_q(runs[k].cls) ? col - runs[k].start : 0);
  }
  /* n characters of the gap-stripped target / query row starting at non-gap index idx */
  std::string ref_slice(bool is_t, uint64_t idx, uint64_t n) const {
    const char* row = is_t ? rec->t().seq_data() : rec->q().seq_data();
    std::string out;
    size_t lo = 0, hi = runs.size(); /* last run whose prefix count is <= idx: it advances */
    while (hi - lo > 1) {
      size_t mid = (lo + hi) / 2;
      if ((is_t ? runs[mid].tb : runs[mid].qb) <= idx)
        lo = mid;
      else
        hi = mid;
    }
    size_t k = lo;
    while (n && k < runs.size()) {
      bool adv = is_t ? adv_t(runs[k].cls) : adv_q(runs[k].cls);
      uint64_t b = is_t ? runs[k].tb : runs[k].qb, len = end(k) - runs[k].start;
      if (adv && idx < b + len) {
        uint64_t take = std::min(n, b + len - idx);
        out.append(row + runs[k].start + (idx - b), take);
        idx += take;
        n -= take;
      }
      k++;
    }
    if (n) fail("panic: VCF REF/ALT slice out of range (caller.rs:500-501,554-555)");
    return out;
  }
};

/* find_safe_chunk_boundary, caller.rs:159-219, on runs: a gap segment = adjacent I/D/W runs */
uint64_t safe_chunk_end(const CallBlock& b, uint64_t start, uint64_t chunk_size, uint64_t svlen) {
  const uint64_t proposed = std::min(start + chunk_size, b.total);
  uint64_t safe_end = proposed;
  size_t k = b.run_at(start);
  while (k < b.runs.size() && b.runs[k].start < proposed) {
    uint32_t c = b.runs[k].cls;
    if (c == 0 || c == 3) {
      k++;
      continue;
    }
    size_t j = k;
    while (j + 1 < b.runs.size() && b.runs[j + 1].cls != 0 && b.runs[j + 1].cls != 3) j++;
    const uint64_t gs = std::max(b.runs[k].start, start), ge = b.end(j);
    if (ge < proposed) {
      if (ge - gs >= svlen) safe_end = ge;
    } else {
      if (proposed - gs >= svlen) safe_end = ge < b.total ? ge : proposed;
    }
    k = j + 1;
  }
  return safe_end;
}

void vcf_line(std::string& out, const std::string& chro, uint64_t pos, const std::string& ref,
              const std::string& alt, bool symbolic, const std::string& info, const std::string& sample) {
  auto bases = [&](const std::string& s) { /* noodles parses bases case-insensitively, prints upper case */
    for (char c : s) {
      char u = (c >= 'a' && c <= 'z') ? (char)(c - 32) : c;
      if (u != 'A' && u != 'C' && u != 'G' && u != 'T' && u != 'N')
        fail(std::string("invalid reference/alternate base `") + c + "` for a VCF record (noodles-vcf parse error)");
      out.push_back(u);
    }
  };
  out += chro;
  out.push_back('\t');
  append_u64(out, pos);
  out += "\t.\t";
  bases(ref);
  out.push_back('\t');
  if (symbolic)
    out += alt;
  else
    bases(alt);
  out += "\t.\t.\t";
  out += info.empty() ? "." : info;
  out += "\tGT:QI\t1|1:";
  out += sample;
  out.push_back('\n');
}

/* call_within_var on chunk [cs, ce), caller.rs:388-608 */
void call_chunk(const CallBlock& b, uint64_t cs, uint64_t ce, bool snp, bool inv, uint64_t svlen, std::string& out) {
  const MafRecord& r = *b.rec;
  const uint64_t tb0 = b.t_before(cs), qb0 = b.q_before(cs);
  const uint64_t t_align = b.t_before(ce) - tb0, q_align = b.q_before(ce) - qb0;
  /* create_chunk_record (:221-265) + accessors (maf.rs:433-450,464-470) */
  const uint64_t t_start = r.t().start + tb0, t_end = t_start + t_align;
  const uint64_t q_sline_start = r.q().start + qb0;
  const bool neg = r.q().neg;
  const uint64_t q_start = neg ? r.q().size - q_sline_start - q_align : q_sline_start;
  const uint64_t q_end = neg ? r.q().size - q_sline_start : q_sline_start + q_align;
  const std::string &chro = r.t().name, &q_chro = r.q().name;
  const char suffix = neg ? 'N' : 'P';
  auto qi = [&](uint64_t a, uint64_t b2, bool three) {
    std::string s = q_chro + "@";
    append_u64(s, a);
    if (!three) {
      s.push_back('@');
      append_u64(s, b2);
    }
    s.push_back('@');
    s.push_back(suffix);
    return s;
  };
  if (neg && t_align != 0 && inv) { /* :423-440 */
    std::string info = "SVTYPE=INV;END=";
    append_u64(info, t_end);
    vcf_line(out, chro, t_start + 1, b.ref_slice(true, tb0, 1), "<INV>", true, info, qi(q_start, q_end, false));
  }
  const std::string init_info = neg ? "INV_NEST=TRUE;" : "";
  uint64_t t_off = t_start, q_off = q_start;
  bool after_m = false;
  for (size_t k = b.run_at(cs); k < b.runs.size() && b.runs[k].start < ce; k++) {
    const uint64_t s0 = std::max(b.runs[k].start, cs), e0 = std::min(b.end(k), ce);
    const uint64_t len = e0 - s0;
    switch (b.runs[k].cls) {
      case 0:
        t_off += len;
        q_off += len;
        after_m = true;
        break;
      case 4: break;
      case 1: /* I :464-515 */
        if (len > svlen) {
          if (!after_m) {
            q_off += len;
            after_m = false;
            continue;
          }
          std::string info = init_info + "SVTYPE=INS;SVLEN=";
          append_u64(info, len);
          info += ";END=";
          append_u64(info, t_off);
          vcf_line(out, chro, t_off, b.ref_slice(true, tb0 + (t_off - t_start - 1), 1),
                   b.ref_slice(false, qb0 + (q_off - q_start - 1), len + 1), false, info, qi(q_off, q_off + len, false));
        }
        q_off += len;
        after_m = false;
        break;
      case 2: /* D :516-569 */
        if (len > svlen) {
          if (!after_m) {
            t_off += len;
            after_m = false;
            continue;
          }
          std::string info = init_info + "SVTYPE=DEL;SVLEN=";
          append_u64(info, len);
          info += ";END=";
          append_u64(info, t_off + len);
          vcf_line(out, chro, t_off, b.ref_slice(true, tb0 + (t_off - t_start - 1), len + 1),
                   b.ref_slice(false, qb0 + (q_off - q_start - 1), 1), false, info, qi(q_off, q_off, false));
        }
        t_off += len;
        after_m = false;
        break;
      case 3: /* X :570-603 */
        if (snp) {
          for (uint64_t x = 0; x < len; x++) {
            vcf_line(out, chro, t_off + 1, b.ref_slice(true, tb0 + (t_off - t_start), 1),
                     b.ref_slice(false, qb0 + (q_off - q_start), 1), false, "", qi(q_off, 0, true));
            t_off++;
            q_off++;
          }
        } else {
          t_off += len;
          q_off += len;
        }
        after_m = true;
        break;
    }
  }
}

/* build_header, caller.rs:304-338 as noodles-vcf 0.43 prints it (README.md:323-331) */
std::string vcf_header(const std::string& sample, const std::vector<std::pair<std::string, uint64_t>>& contigs) {
  std::string h =
      "##fileformat=VCFv4.4\n"
      "##INFO=<ID=SVLEN,Number=A,Type=Integer,Description=\"Length of structural variant\">\n"
      "##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"Type of structural variant\">\n"
      "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the longest variant described in this record\">\n"
      "##INFO=<ID=INV_NEST,Number=1,Type=String,Description=\"Varations nested within inversion\">\n"
      "##FORMAT=<ID=QI,Number=1,Type=String,Description=\"Query informations\">\n"
      "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n";
  for (auto& c : contigs) { /* add_header_contig :340-357; placement after FORMAT is unpinned */
    h += "##contig=<ID=" + c.first + ",length=";
    append_u64(h, c.second);
    h += ">\n";
  }
  h += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample + "\n";
  return h;
}

/* ---- call (PAF) (caller.rs:268-302, 610-822) -----------------------------------------------------------
 * GPU: the op walk (wga_paf_call_events) and the event -> VCF row text (wga_paf_call_vcf).  Host: fetch coordinates. */
/* the records which[0 .. n_which) of a piece (which == nullptr: all of them) in resident batches; their VCF rows are
 * appended to `body` in that order (`sizes`, when given, gets every record's byte count).  Errors are thrown; *bad_at is then
 * the position (in `which` order) of the record they belong to. */
void call_paf_run(Dev& d, DevFasta& tf, DevFasta& qf, const PafInput& pin, const size_t* which, size_t n_which,
                  const uint8_t* d_text_here, bool snp, uint64_t svlen, std::string& body, std::vector<uint64_t>* sizes,
                  size_t* bad_at) {
  const std::vector<PafRecord>& recs = pin.recs;
  const size_t keep = d.owned.size(); /* the input text, the pools */
  const uint64_t kMaxText = 160ull << 20; /* ~64 M ops per batch */
  size_t i0 = 0;
  while (i0 < n_which) {
    std::vector<uint64_t> t_off, t_len, q_off, q_len;
    size_t i = i0;
    uint64_t est_text = 0;
    for (; i < n_which; i++) {
      const size_t ri = which ? which[i] : i;
      const PafRecord& r = recs[ri];
      *bad_at = i; /* an error raised from here on belongs to this record */
      if (i > i0 && est_text > kMaxText) break;
      uint64_t to, tl, qo, ql;
      tf.fetch(r.target_name, r.target_start, r.target_end, &to, &tl); /* paf.rs:221-237: end inclusive */
      qf.fetch(r.query_name, r.query_start, r.query_end, &qo, &ql);
      if (r.neg && tl == 0) fail("panic: byte index 1 is out of bounds of the fetched target (caller.rs:642)");
      { /* a missing tag or an empty CIGAR ends the run at this record (checked here to keep the order of the errors) */
        bool has_tag, empty;
        if (pin.on_device) {
          has_tag = pin.cg_beg[ri] != WGA_NONE;
          empty = has_tag && pin.cg_end[ri] == pin.cg_beg[ri];
        } else {
          int err = 0;
          const std::string cg = paf_cigar_string(r, &err);
          has_tag = !err;
          empty = has_tag && cg.size() == 5;
        }
        if (!has_tag) fail("CIGAR start tag not found");
        if (empty) fail(cigar_error_message(WGA_REC_PANIC, std::string(), 0, 0));
      }
      est_text += pin.cigar_bytes(ri);
      t_off.push_back(to);
      t_len.push_back(tl);
      q_off.push_back(qo);
      q_len.push_back(ql);
    }
    /* the CIGARs are tokenised on the device; tokeniser errors end a record's walk but are discarded
     * (:673,815-819): the ops in front of them are kept.  Only a missing tag and an empty CIGAR are fatal. */
    CigarTexts cigars;
    wga_cigar_batch cb;
    std::vector<wga_tok_err> terrs;
    const uint32_t n_asked = (uint32_t)(i - i0);
    const std::string tag_err = device_tokenise(d, pin, i0, n_asked, cigars, &cb, &terrs, which, d_text_here);
    for (uint32_t k = 0; k < cb.n; k++)
      if (terrs[k].err == WGA_REC_PANIC) {
        *bad_at = i0 + k;
        fail(cigar_error_message(WGA_REC_PANIC, std::string(), 0, 0));
      }
    *bad_at = i0 + cb.n;
    if (cb.n < n_asked) fail(tag_err); /* records are processed in order: the first failing one ends the run */
    const uint32_t n = cb.n;
    if (n) {
      auto* d_cnt = (uint64_t*)d.alloc((size_t)n * 8);
      d.check(wga_paf_call_events(d.ctx, &cb, svlen, snp, d_cnt, nullptr, nullptr));
      auto* d_eoff = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
      d.check(wga_exclusive_scan_u64(d.ctx, n, d_cnt, d_eoff));
      uint64_t n_ev = 0;
      d.download(&n_ev, (const uint64_t*)d_eoff + n, 1);
      auto* d_ev = (uint64_t*)d.alloc((3 * n_ev + 3) * 8);
      d.check(wga_paf_call_events(d.ctx, &cb, svlen, snp, d_cnt, d_ev, d_eoff));
      /* the rows are formatted where the events lie (wga_paf_call_vcf): names, coordinates and the places of the fetched
       * sequences go up, the text comes back; the driver is buffered, so an error in any record leaves nothing written
       * (:294-299) */
      std::string names;
      std::unordered_map<std::string, uint64_t> name_at;
      auto name_off = [&](const std::string& nm) {
        auto it = name_at.find(nm);
        if (it != name_at.end()) return it->second;
        const uint64_t at = names.size();
        names += nm;
        name_at.emplace(nm, at);
        return at;
      };
      std::vector<wga_vcf_rec> vr(n);
      for (uint32_t k = 0; k < n; k++) {
        const PafRecord& r = recs[which ? which[i0 + k] : i0 + k];
        wga_vcf_rec& v = vr[k];
        v.t_name_off = name_off(r.target_name), v.t_name_len = (uint32_t)r.target_name.size();
        v.q_name_off = name_off(r.query_name), v.q_name_len = (uint32_t)r.query_name.size();
        v.t_start = r.target_start, v.t_end = r.target_end, v.q_start = r.query_start, v.q_end = r.query_end;
        v.t_off = t_off[k], v.t_len = t_len[k], v.q_off = q_off[k], v.q_len = q_len[k];
      }
      names.push_back('\0'); /* never empty */
      const uint8_t* d_names = d.upload((const uint8_t*)names.data(), names.size());
      const wga_vcf_rec* d_vr = d.upload(vr.data(), vr.size());
      auto* d_nb = (uint64_t*)d.alloc((size_t)n * 8);
      auto* d_err = (wga_vcf_err*)d.alloc((size_t)n * sizeof(wga_vcf_err));
      d.check(wga_paf_call_vcf(d.ctx, &cb, svlen, d_ev, d_eoff, d_vr, d_names, tf.d_pool, qf.d_pool, d_nb, d_err, nullptr,
                               nullptr));
      auto* d_toff = (uint64_t*)d.alloc(((size_t)n + 1) * 8);
      d.check(wga_exclusive_scan_u64(d.ctx, n, d_nb, d_toff));
      std::vector<wga_vcf_err> errs(n);
      d.download(errs.data(), (const wga_vcf_err*)d_err, n);
      for (uint32_t k = 0; k < n; k++) { /* the first failing record in input order */
        if (errs[k].item == WGA_NONE) continue;
        *bad_at = i0 + k;
        if (errs[k].kind == 1)
          fail("panic: VCF REF/ALT slice out of the fetched sequence (caller.rs:695-696,753-754,800-801)");
        fail(std::string("invalid reference/alternate base `") + (char)errs[k].ch +
             "` for a VCF record (noodles-vcf parse error)");
      }
      uint64_t n_text = 0;
      d.download(&n_text, (const uint64_t*)d_toff + n, 1);
      if (sizes) {
        std::vector<uint64_t> nb(n);
        d.download(nb.data(), (const uint64_t*)d_nb, n);
        sizes->insert(sizes->end(), nb.begin(), nb.end());
      }
      if (n_text) {
        auto* d_text = (uint8_t*)d.alloc(n_text + 64);
        d.check(wga_paf_call_vcf(d.ctx, &cb, svlen, d_ev, d_eoff, d_vr, d_names, tf.d_pool, qf.d_pool, nullptr, nullptr,
                                 d_text, d_toff));
        const size_t at = body.size();
        body.resize(at + n_text);
        d.download((uint8_t*)&body[at], (const uint8_t*)d_text, n_text);
      }
    }
    d.release_to(keep);
    i0 = i;
  }
  *bad_at = n_which;
}

int cmd_call_paf_multi(const std::string* input, const std::string& t_fa, const std::string& q_fa, bool snp, uint64_t svlen,
                       const std::string& sample, Output& out, int ngpu) {
  /* --gpus N: the records go to the devices by fnv1a64(target_name) % N; every device walks and formats its share, the rows
   * meet on the host in input order (the driver is buffered: caller.rs:294-299); the first failing record in input order
   * decides the message */
  std::vector<std::unique_ptr<Dev>> devs;
  for (int g = 0; g < ngpu; g++) devs.emplace_back(new Dev(g));
  PafInput pin = load_paf(*devs[0], input, false);
  const size_t n = pin.recs.size();
  std::vector<DevFasta> tf(ngpu), qf(ngpu);
  std::vector<std::vector<size_t>> mine(ngpu);
  for (size_t i = 0; i < n; i++) mine[fnv1a64(pin.recs[i].target_name) % (uint64_t)ngpu].push_back(i);
  std::string text16;
  if (pin.on_device && ngpu > 1) text16 = pin.text + std::string(16, '\0');
  std::vector<std::string> bodies(ngpu), msgs(ngpu);
  std::vector<std::vector<uint64_t>> sizes(ngpu);
  std::vector<size_t> bad(ngpu, n); /* input index of a worker's failing record */
  on_devices(ngpu, [&](int g) {
    Dev& d = *devs[g];
    size_t at = 0;
    try {
      d.init();
      tf[g].load(d, t_fa);
      qf[g].load(d, q_fa);
      const uint8_t* d_text = nullptr;
      if (pin.on_device) d_text = g == 0 ? pin.d_text : d.upload((const uint8_t*)text16.data(), text16.size());
      call_paf_run(d, tf[g], qf[g], pin, mine[g].data(), mine[g].size(), d_text, snp, svlen, bodies[g], &sizes[g], &at);
    } catch (Error& e) {
      bad[g] = at < mine[g].size() ? mine[g][at] : (mine[g].empty() ? 0 : mine[g].back());
      msgs[g] = e.msg.empty() ? std::string("error") : e.msg;
    } catch (std::exception& e) { /* bad_alloc and friends: an exception leaving a thread is std::terminate */
      bad[g] = at < mine[g].size() ? mine[g][at] : 0;
      msgs[g] = std::string("internal error: ") + e.what();
    }
  });
  int first = -1;
  for (int g = 0; g < ngpu; g++)
    if (!msgs[g].empty() && (first < 0 || bad[g] < bad[first])) first = g;
  if (first >= 0) fail(msgs[first]);
  std::string body;
  {
    size_t total = 0;
    for (int g = 0; g < ngpu; g++) total += bodies[g].size();
    body.reserve(total);
    std::vector<size_t> next(ngpu, 0), pos(ngpu, 0);
    for (size_t i = 0; i < n; i++) {
      const int g = (int)(fnv1a64(pin.recs[i].target_name) % (uint64_t)ngpu);
      const uint64_t sz = sizes[g][next[g]++];
      body.append(bodies[g], pos[g], sz);
      pos[g] += sz;
    }
  }
  out.write(vcf_header(sample, {}));
  out.write(body);
  out.close();
  return 0;
}

int cmd_call_paf(const std::string* input, const std::string& t_fa, const std::string& q_fa, bool snp,
                 uint64_t svlen, const std::string& sample, Output& out) {
  if (g_gpus > 1) return cmd_call_paf_multi(input, t_fa, q_fa, snp, svlen, sample, out, g_gpus);
  Dev d;
  PafInput pin = load_paf(d, input, false);
  DevFasta tf, qf;
  d.init();
  tf.load(d, t_fa);
  qf.load(d, q_fa);
  std::string body;
  size_t bad_at = 0;
  call_paf_run(d, tf, qf, pin, nullptr, pin.recs.size(), nullptr, snp, svlen, body, nullptr, &bad_at);
  /* everything is buffered; the header goes out first, after all records were processed (:294-299) */
  out.write(vcf_header(sample, {}));
  out.write(body);
  out.close();
  return 0;
}

int cmd_call_maf(const std::string* input, bool snp, bool inv, uint64_t svlen, const std::string& sample,
                 const std::string* query_name, const std::string* query_regex, uint64_t chunk_size, Output& out) {
  Dev d;
  MafDevices md(d);
  MafChunks chunks(input);
  /* utils.rs:414-436: `<input>.index` (JSON written by `maf-index`) supplies ##contig lines */
  std::vector<std::pair<std::string, uint64_t>> contigs;
  if (input) contigs = maf_index_ref_contigs(*input + ".index");
  std::string text = vcf_header(sample, contigs);
  std::regex re;
  if (!query_name && query_regex) { /* cli.rs:332-343 anchors the pattern; maf.rs:267-271 searches from 0 */
    std::string pat = *query_regex;
    if (pat.empty() || pat.front() != '^') pat.insert(pat.begin(), '^');
    if (pat.back() != '$') pat.push_back('$');
    re = std::regex(pat);
  }
  if (chunk_size == 0) fail("chunk size must be positive (the reference would never terminate)");
  MafInput min;
  std::vector<uint64_t> runs_keep; /* one device: the run list's memory serves every piece (hundreds of megabytes; fresh pages cost more than the copy) */
  g_timer.mark("host");
  while (chunks.next(d, min)) { /* one piece of the file at a time, rows written as they are called (caller.rs:62-149) */
  g_timer.mark("read + upload + split");
  std::vector<MafRecord>& all = min.recs;
  /* record selection (:62-108): single-s-line blocks and blocks without the asked query are skipped */
  std::vector<const MafRecord*> recs;
  for (auto& r : all) {
    if (r.slines.size() == 1) continue;
    if (query_name) {
      size_t k = 0;
      for (; k < r.slines.size(); k++)
        if (r.slines[k].name == *query_name) break;
      if (k == r.slines.size()) continue;
      r.query_idx = k;
    } else if (query_regex) {
      size_t k = 0;
      for (; k < r.slines.size(); k++)
        if (std::regex_search(r.slines[k].name, re)) break;
      if (k == r.slines.size()) continue;
      r.query_idx = k;
    } else {
      r.query_idx = 1;
    }
    if (r.q().seq_size() < r.t().seq_size())
      fail("panic: query row shorter than the target row (caller.rs:175 slice out of range)");
    recs.push_back(&r);
  }
  const uint32_t n = (uint32_t)recs.size();
  if (n) {
    /* every device walks its range of the blocks (K4: count, scan, fill); the run lists meet here in block order */
    std::vector<uint64_t> roff(n + 1, 0), cols(n), cnt_all(n);
    std::vector<std::vector<uint64_t>> runs_of(md.count());
    if (md.count() == 1) runs_of[0] = std::move(runs_keep);
    std::vector<uint32_t> lo_of(md.count(), 0), n_of(md.count(), 0);
    md.run(min, recs, true /* total_size = target row length (:115) */, [&](int g, Dev& dg, const MafRows& p, uint32_t lo, uint32_t cnt) {
      auto* d_cnt = (uint64_t*)dg.alloc((size_t)cnt * 8);
      dg.check(wga_maf_call_runs(dg.ctx, cnt, p.d_rows, p.d_t, p.d_q, p.d_c, d_cnt, nullptr, nullptr));
      auto* d_roff = (uint64_t*)dg.alloc(((size_t)cnt + 1) * 8);
      dg.check(wga_exclusive_scan_u64(dg.ctx, cnt, d_cnt, d_roff));
      std::vector<uint64_t> ro(cnt + 1);
      dg.download(ro.data(), d_roff, cnt + 1);
      auto* d_runs = (uint64_t*)dg.alloc((3 * ro[cnt] + 3) * 8);
      dg.check(wga_maf_call_runs(dg.ctx, cnt, p.d_rows, p.d_t, p.d_q, p.d_c, d_cnt, d_runs, d_roff));
      runs_of[g].resize(3 * ro[cnt]);
      if (ro[cnt]) dg.download(runs_of[g].data(), d_runs, 3 * ro[cnt]);
      for (uint32_t k = 0; k < cnt; k++) cnt_all[lo + k] = ro[k + 1] - ro[k], cols[lo + k] = p.cols[k];
      lo_of[g] = lo, n_of[g] = cnt;
    });
    for (uint32_t k = 0; k < n; k++) roff[k + 1] = roff[k] + cnt_all[k];
    std::vector<uint64_t> runs;
    if (md.count() == 1 && n_of[0] == n && lo_of[0] == 0) { /* one device: its list is the list (hundreds of megabytes per piece) */
      runs = std::move(runs_of[0]);
    } else {
      runs.resize(3 * roff[n]);
      for (int g = 0; g < md.count(); g++)
        if (n_of[g]) std::copy(runs_of[g].begin(), runs_of[g].end(), runs.begin() + 3 * roff[lo_of[g]]);
    }
    runs_of.clear();
    g_timer.mark("kernels + run list download");
    /* the event rules and the VCF text of a block depend on that block alone: contiguous ranges of blocks go to host
     * threads, their text is written in block order (rows in front of a failing block are written, then the error) */
    unsigned nthr = std::thread::hardware_concurrency();
    nthr = std::max(1u, std::min({nthr, 96u, n / 256u + 1u})); /* rows are short strings: the work scales with the cores */
    if (const char* e = getenv("WGA_HOST_THREADS")) nthr = std::max(1u, std::min((unsigned)atoi(e), n)); /* tests */
    std::vector<std::string> parts(nthr), errs(nthr);
    auto work = [&](unsigned t) {
      const uint32_t lo = (uint32_t)((uint64_t)n * t / nthr), hi = (uint32_t)((uint64_t)n * (t + 1) / nthr);
      std::string& txt = parts[t];
      try {
        CallBlock b;
        for (uint32_t k = lo; k < hi; k++) {
          b.rec = recs[k];
          b.total = cols[k];
          b.runs.clear();
          b.runs.reserve(roff[k + 1] - roff[k]);
          for (uint64_t x = roff[k]; x < roff[k + 1]; x++)
            b.runs.push_back(CallRun{runs[3 * x] >> 3, runs[3 * x + 1], runs[3 * x + 2], (uint32_t)(runs[3 * x] & 7)});
          uint64_t cs = 0;
          while (cs < b.total) {
            uint64_t ce = safe_chunk_end(b, cs, chunk_size, svlen);
            call_chunk(b, cs, ce, snp, inv, svlen, txt);
            if (ce <= cs) fail("panic: chunk boundary did not advance");
            cs = ce;
          }
        }
      } catch (Error& e) {
        errs[t] = e.msg.empty() ? std::string("error") : e.msg;
      } catch (std::exception& e) {
        errs[t] = std::string("internal error: ") + e.what();
      }
    };
    {
      std::vector<std::thread> th;
      for (unsigned t = 1; t < nthr; t++) th.emplace_back(work, t);
      work(0);
      for (auto& x : th) x.join();
    }
    g_timer.mark("host event rules + VCF text");
    {
      /* the threads' texts leave in block order: what was collected before, then part 0, 1 ...; up to the first part that
       * holds an error (its rows in front of the failing block are written, then the error is raised).  Into a plain file
       * the parts are written side by side at their places (no copy into one string, no single writer). */
      unsigned good = 0;
      while (good < nthr && errs[good].empty()) good++;
      const unsigned upto = good < nthr ? good + 1 : nthr; /* parts written */
      out.write(text);
      text.clear();
      uint64_t pos0 = 0;
      const int fd = nthr > 1 ? out.plain_fd(&pos0) : -1;
      if (fd >= 0) {
        std::vector<uint64_t> at(upto + 1, pos0);
        for (unsigned t = 0; t < upto; t++) at[t + 1] = at[t] + parts[t].size();
        std::atomic<bool> bad(false);
        auto put = [&](unsigned t) {
          size_t w = 0;
          while (w < parts[t].size()) {
            const ssize_t r2 = pwrite(fd, parts[t].data() + w, parts[t].size() - w, (off_t)(at[t] + w));
            if (r2 <= 0) {
              bad = true;
              return;
            }
            w += (size_t)r2;
          }
        };
        std::vector<std::thread> th;
        const unsigned nw = std::min(upto, 8u);
        for (unsigned k = 1; k < nw; k++)
          th.emplace_back([&, k] {
            for (unsigned t = k; t < upto; t += nw) put(t);
          });
        for (unsigned t = 0; t < upto; t += nw) put(t);
        for (auto& x : th) x.join();
        if (bad) fail("write error on the output file");
        out.advance(at[upto] - pos0);
      } else {
        for (unsigned t = 0; t < upto; t++) out.write(parts[t]);
      }
      if (good < nthr) fail(errs[good]);
    }
    g_timer.mark("write");
    if (md.count() == 1) runs_keep = std::move(runs);
  }
  md.release_all();
  }
  out.write(text);
  out.close();
  return 0;
}

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
