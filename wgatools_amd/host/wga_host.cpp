/*
 * wga_host.cpp — parsers / writers of the host layer (see wga_host.hpp for the reference map).
 */
#include "wga_host.hpp"

#include <ctype.h>
#include <stdio.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <thread>
#include <cmath>

namespace wga {

void fail(const std::string& msg) { throw Error{msg}; }

/* ------------------------------------------------------------------------------------------ */
/* I/O (utils.rs:135-246)                                                                      */
/* ------------------------------------------------------------------------------------------ */
std::string read_all(const std::string* path) {
  std::string out;
  if (!path) {
    if (isatty(0)) fail("Empty stdin, please add `-h` for help"); /* errors.rs:23 */
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, stdin)) > 0) out.append(buf, n);
    return out;
  }
  { /* a plain file: one read (zlib's transparent mode copies at ~1.2 GB/s) */
    FILE* pf = fopen(path->c_str(), "rb");
    if (!pf) fail("File path `" + *path + "` not exist"); /* errors.rs:13 */
    unsigned char m[2] = {0, 0};
    const size_t got = fread(m, 1, 2, pf);
    if (!(got == 2 && m[0] == 31 && m[1] == 139)) {
      struct stat st;
      if (fstat(fileno(pf), &st) == 0 && S_ISREG(st.st_mode)) {
        out.resize((size_t)st.st_size);
        fseek(pf, 0, SEEK_SET);
        const size_t rd = out.empty() ? 0 : fread(&out[0], 1, out.size(), pf);
        fclose(pf);
        if (rd != out.size()) fail("IO error:short read of `" + *path + "`");
        return out;
      }
    }
    fclose(pf);
  }
  /* gzopen inflates gzip members (MultiGzDecoder) and reads anything else transparently */
  gzFile f = gzopen(path->c_str(), "rb");
  if (!f) fail("File path `" + *path + "` not exist"); /* errors.rs:13 */
  gzbuffer(f, 1 << 20);
  char buf[1 << 16];
  int n;
  while ((n = gzread(f, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
  if (n < 0) { /* a damaged / truncated gzip member: the reference's decoder raises an io::Error (errors.rs:10) */
    int ec = 0;
    const std::string why = gzerror(f, &ec);
    gzclose(f);
    fail("IO error:" + why);
  }
  gzclose(f);
  return out;
}

/* ---- BGZF: gzip members of <= 64 KB whose extra field carries the block size ("BC", SAM spec 4.1) ---------------- */
namespace {
struct BgzfBlock {
  size_t cdata, clen; /* the raw deflate stream inside the file image */
  size_t out, isize;
};
inline uint32_t le16(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t le32(const unsigned char* p) { return le16(p) | (le16(p + 2) << 16); }
/* walks the members; false if the file is not BGZF from end to end */
bool bgzf_blocks(const std::string& img, std::vector<BgzfBlock>& blocks, size_t* total) {
  const unsigned char* b = (const unsigned char*)img.data();
  size_t p = 0, n = img.size(), out = 0;
  while (p < n) {
    if (n - p < 18 || b[p] != 31 || b[p + 1] != 139 || b[p + 2] != 8 || !(b[p + 3] & 4)) return false;
    const size_t xlen = le16(b + p + 10);
    if (n - p < 12 + xlen) return false;
    size_t bsize = 0;
    for (size_t x = p + 12; x + 4 <= p + 12 + xlen;) {
      const size_t slen = le16(b + x + 2);
      if (b[x] == 'B' && b[x + 1] == 'C' && slen == 2 && x + 6 <= p + 12 + xlen) bsize = le16(b + x + 4) + 1u;
      x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || bsize > n - p) return false;
    BgzfBlock k;
    k.cdata = p + 12 + xlen;
    k.clen = bsize - (12 + xlen) - 8;
    k.isize = le32(b + p + bsize - 4);
    k.out = out;
    out += k.isize;
    blocks.push_back(k);
    p += bsize;
  }
  *total = out;
  return !blocks.empty();
}
}  // namespace

uint32_t gzip_crc32(const void* p, size_t n) { return (uint32_t)crc32(0L, (const Bytef*)p, (uInt)n); }

bool read_bgzf_image(const std::string& path, std::string& img, std::vector<BgzfMember>& members, uint64_t* total) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) fail("File path `" + path + "` not exist"); /* errors.rs:13 */
  unsigned char magic[4] = {0, 0, 0, 0};
  const size_t got = fread(magic, 1, 4, f);
  if (!(got == 4 && magic[0] == 31 && magic[1] == 139 && magic[2] == 8 && (magic[3] & 4))) {
    fclose(f);
    return false;
  }
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  img.resize(sz > 0 ? (size_t)sz : 0);
  const bool ok = img.empty() || fread(&img[0], 1, img.size(), f) == img.size();
  fclose(f);
  if (!ok) fail("IO error:short read of `" + path + "`");
  std::vector<BgzfBlock> blocks;
  size_t out = 0;
  if (!bgzf_blocks(img, blocks, &out)) {
    img.clear();
    return false;
  }
  members.clear();
  for (const BgzfBlock& k : blocks) {
    if (k.isize == 0) continue; /* the EOF marker */
    if (k.clen > 0xFFFFFFFFull || k.isize > 0xFFFFFFFFull) return false;
    members.push_back(BgzfMember{(uint64_t)k.cdata, (uint32_t)k.clen, (uint32_t)k.isize, (uint64_t)k.out});
  }
  *total = (uint64_t)out;
  return true;
}

/* The member table of a BGZF file without the file in memory: every member's header names its size ("BC"), so the walk hops
 * from header to header — one pread of the trailer in front (CRC-32, ISIZE) and the next header per member. */
bool scan_bgzf_members(const std::string& path, std::vector<BgzfMember>& members, uint64_t* total, uint64_t* file_bytes) {
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) fail("File path `" + path + "` not exist"); /* errors.rs:13 */
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
    ::close(fd);
    return false;
  }
  const uint64_t n = (uint64_t)st.st_size;
  members.clear();
  uint64_t p = 0, out = 0;
  bool ok = n > 0;
  std::vector<unsigned char> hdr(96);
  while (ok && p < n) {
    if (n - p < 18) {
      ok = false;
      break;
    }
    size_t want = (size_t)std::min<uint64_t>(hdr.size(), n - p);
    if (pread(fd, hdr.data(), want, (off_t)p) != (ssize_t)want) {
      ok = false;
      break;
    }
    const unsigned char* b = hdr.data();
    if (b[0] != 31 || b[1] != 139 || b[2] != 8 || !(b[3] & 4)) {
      ok = false;
      break;
    }
    const size_t xlen = le16(b + 10);
    if (n - p < 12 + xlen) {
      ok = false;
      break;
    }
    if (12 + xlen > want) { /* an extra field longer than the usual six bytes */
      hdr.resize(12 + xlen);
      want = 12 + xlen;
      if (pread(fd, hdr.data(), want, (off_t)p) != (ssize_t)want) {
        ok = false;
        break;
      }
      b = hdr.data();
    }
    size_t bsize = 0;
    for (size_t x = 12; x + 4 <= 12 + xlen;) {
      const size_t slen = le16(b + x + 2);
      if (b[x] == 'B' && b[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = le16(b + x + 4) + 1u;
      x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || bsize > n - p) {
      ok = false;
      break;
    }
    unsigned char tr[4];
    if (pread(fd, tr, 4, (off_t)(p + bsize - 4)) != 4) {
      ok = false;
      break;
    }
    const uint64_t isize = le32(tr);
    if (isize) members.push_back(BgzfMember{p + 12 + xlen, (uint32_t)(bsize - (12 + xlen) - 8), (uint32_t)isize, out}); /* not the EOF marker */
    out += isize;
    p += bsize;
  }
  ::close(fd);
  if (!ok) {
    members.clear();
    return false;
  }
  *total = out;
  if (file_bytes) *file_bytes = n;
  return true;
}

std::string read_all_parallel(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) fail("File path `" + path + "` not exist"); /* errors.rs:13 */
  unsigned char magic[4] = {0, 0, 0, 0};
  const size_t got = fread(magic, 1, 4, f);
  const bool maybe_bgzf = got == 4 && magic[0] == 31 && magic[1] == 139 && magic[2] == 8 && (magic[3] & 4);
  if (!maybe_bgzf) {
    fclose(f);
    return read_all(&path); /* plain text or ordinary gzip */
  }
  std::string img;
  {
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    img.resize(sz > 0 ? (size_t)sz : 0);
    if (!img.empty() && fread(&img[0], 1, img.size(), f) != img.size()) {
      fclose(f);
      fail("IO error:short read of `" + path + "`");
    }
    fclose(f);
  }
  std::vector<BgzfBlock> blocks;
  size_t total = 0;
  if (!bgzf_blocks(img, blocks, &total)) return read_all(&path);
  std::string out(total, '\0');
  std::atomic<size_t> next{0};
  std::atomic<int> bad{0};
  auto work = [&]() {
    z_stream zs;
    for (;;) {
      const size_t a = next.fetch_add(16);
      if (a >= blocks.size()) break;
      for (size_t i = a; i < std::min(blocks.size(), a + 16); i++) {
        const BgzfBlock& k = blocks[i];
        if (k.isize == 0) continue; /* the EOF marker block */
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) {
          bad = 1;
          continue;
        }
        zs.next_in = (Bytef*)(img.data() + k.cdata);
        zs.avail_in = (uInt)k.clen;
        zs.next_out = (Bytef*)&out[k.out];
        zs.avail_out = (uInt)k.isize;
        const int rc = inflate(&zs, Z_FINISH);
        if (rc != Z_STREAM_END || zs.avail_out != 0) bad = 1;
        inflateEnd(&zs);
      }
    }
  };
  unsigned nthr = std::thread::hardware_concurrency();
  nthr = nthr < 1 ? 1 : (nthr > 32 ? 32 : nthr);
  if (blocks.size() < 64) nthr = 1;
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nthr; t++) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  if (bad) fail("IO error:corrupt BGZF block in `" + path + "`");
  return out;
}

void LineChunkReader::open(const std::string* path) {
  if (source) return; /* the caller's producer stands for the file */
  if (!path) {
    if (isatty(0)) fail("Empty stdin, please add `-h` for help"); /* errors.rs:23 */
    is_stdin = true;
    return;
  }
  { /* gzip magic? (utils.rs:135-189 sniffs the first bytes the same way) */
    FILE* f = fopen(path->c_str(), "rb");
    if (!f) fail("File path `" + *path + "` not exist"); /* errors.rs:13 */
    unsigned char m[2] = {0, 0};
    const size_t got = fread(m, 1, 2, f);
    fclose(f);
    if (!(got == 2 && m[0] == 31 && m[1] == 139)) {
      fd = ::open(path->c_str(), O_RDONLY);
      if (fd < 0) fail("File path `" + *path + "` not exist");
      struct stat st;
      if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) file_left = (uint64_t)st.st_size;
      return;
    }
  }
  gz = gzopen(path->c_str(), "rb");
  if (!gz) fail("File path `" + *path + "` not exist"); /* errors.rs:13 */
  gzbuffer((gzFile)gz, 1 << 20);
}
LineChunkReader::~LineChunkReader() {
  if (gz) gzclose((gzFile)gz);
  if (fd >= 0) ::close(fd);
}
void LineChunkReader::recycle(std::string&& piece) {
  if (piece.capacity() < ((size_t)8 << 20)) return;
  std::lock_guard<std::mutex> g(spare_mx);
  if (spares.size() < 3) spares.push_back(std::move(piece));
}
/* work on [0, n) shared out over a few threads (pieces of hundreds of megabytes: one core counts line ends at ~5 GB/s) */
template <class F>
static void in_slices(size_t n, F&& f) {
  const unsigned T = n >= ((size_t)32 << 20) ? 8u : 1u;
  if (T == 1u) {
    f(0u, (size_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  for (unsigned t = 1; t < T; t++) th.emplace_back([&f, n, t, T] { f(t, n * t / T, n * (t + 1) / T); });
  f(0u, (size_t)0, n / T);
  for (auto& x : th) x.join();
}
bool LineChunkReader::next(std::string& piece, size_t target, size_t keep) {
  /* piece[0, keep) is the caller's own prefix (kept, not counted); the carried-over tail and fresh reads follow */
  piece.resize(keep);
  bytes_before = next_bytes;
  lines_before = next_lines;
  if (target == 0) target = 1;
  const size_t kRead = (size_t)8 << 20;
  if (fd >= 0 && file_left) { /* one allocation for the piece: what is left of the file, or the target plus a long line */
    const size_t need = keep + carry.size() + (size_t)std::min<uint64_t>(file_left, (uint64_t)target + 2 * kRead) + kRead + 64;
    if (piece.capacity() < need) { /* a buffer that has been through here before, if one is large enough */
      std::lock_guard<std::mutex> g(spare_mx);
      for (size_t k = 0; k < spares.size(); k++)
        if (spares[k].capacity() >= need) {
          spares[k].assign(piece.data(), keep);
          piece.swap(spares[k]);
          spares.erase(spares.begin() + (long)k);
          break;
        }
    }
    if (piece.capacity() < need) {
      piece.reserve(need);
      if (piece.capacity() >= ((size_t)8 << 20)) { /* fewer first-touch faults where transparent huge pages are on request */
        const uintptr_t a = ((uintptr_t)piece.data() + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
        const uintptr_t z = ((uintptr_t)piece.data() + piece.capacity()) & ~(((uintptr_t)2 << 20) - 1);
        if (z > a) madvise((void*)a, (size_t)(z - a), MADV_HUGEPAGE);
      }
    }
  }
  piece += carry;
  carry.clear();
  auto more = [&]() -> bool { /* reads straight into the end of `piece` */
    if (eof) return false;
    const size_t at = piece.size();
    piece.resize(at + kRead);
    size_t n, kept_size = 0;
    if (source) {
      n = source(&piece[at], kRead);
    } else if (is_stdin) {
      n = fread(&piece[at], 1, kRead, stdin);
    } else if (fd >= 0 && file_left > 4 * kRead && target > 4 * kRead) {
      /* a large regular file: what this piece still needs is read by a few threads at once (pread into disjoint
       * ranges of the buffer): one reader moves ~1.8 GB/s out of the page cache */
      const size_t have = at - keep;
      const size_t want = (size_t)std::min<uint64_t>(file_left, (uint64_t)(have < target ? target - have : 0) + kRead);
      piece.resize(at + want);
      const off_t pos = lseek(fd, 0, SEEK_CUR);
      const unsigned T = 8;
      std::vector<size_t> got(T, 0);
      std::vector<std::thread> th;
      auto rd = [&](unsigned t) {
        const size_t a = want * t / T, z = want * (t + 1) / T;
        size_t done = a;
        while (done < z) {
          const ssize_t r = pread(fd, &piece[at + done], z - done, pos + (off_t)done);
          if (r <= 0) break;
          done += (size_t)r;
        }
        got[t] = done - a;
      };
      for (unsigned t = 1; t < T; t++) th.emplace_back(rd, t);
      rd(0);
      for (auto& x : th) x.join();
      n = 0;
      for (unsigned t = 0; t < T; t++) {
        if (got[t] != want * (t + 1) / T - want * t / T) fail("IO error:short read");
        n += got[t];
      }
      lseek(fd, pos + (off_t)n, SEEK_SET);
      file_left -= (uint64_t)n;
      kept_size = at + n;
    } else if (fd >= 0) {
      const ssize_t got = ::read(fd, &piece[at], kRead);
      if (got < 0) fail("IO error:read failed");
      n = (size_t)got;
      file_left -= std::min<uint64_t>(file_left, (uint64_t)n);
    } else {
      const int got = gzread((gzFile)gz, &piece[at], (unsigned)kRead);
      if (got < 0) { /* a damaged / truncated gzip member is an error, not the end of the input (errors.rs:10) */
        int ec = 0;
        fail(std::string("IO error:") + gzerror((gzFile)gz, &ec));
      }
      n = (size_t)got;
    }
    piece.resize(kept_size ? kept_size : at + n);
    if (n == 0) eof = true;
    return n != 0;
  };
  size_t scanned = keep; /* bytes of `piece` already searched for a quote */
  bool whole = false;
  for (;;) {
    if (!whole) { /* a quote anywhere in what came in? */
      std::atomic<bool> quote{false};
      const char* const base = piece.data() + scanned;
      in_slices(piece.size() - scanned, [&](unsigned, size_t a, size_t z) {
        if (z > a && memchr(base + a, '"', z - a)) quote.store(true, std::memory_order_relaxed);
      });
      whole = quote.load();
    }
    scanned = piece.size();
    if (whole) { /* a quoted field may hold line ends: no cut is safe, take the rest of the input */
      while (more()) {
      }
      break;
    }
    if (piece.size() - keep >= target) { /* the piece ends behind the first line end at or after `target` bytes */
      size_t cut = piece.find('\n', keep + target - 1);
      if (cut != std::string::npos) {
        carry.assign(piece, cut + 1, std::string::npos);
        piece.resize(cut + 1);
        break;
      }
    }
    if (!more()) break;
  }
  if (piece.size() == keep) return false;
  next_bytes = bytes_before + (piece.size() - keep);
  {
    uint64_t part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const char* const base = piece.data() + keep;
    in_slices(piece.size() - keep, [&](unsigned t, size_t a, size_t z) { part[t] = (uint64_t)std::count(base + a, base + z, '\n'); });
    next_lines = lines_before;
    for (uint64_t v : part) next_lines += v;
  }
  return true;
}

static bool ends_with(const std::string& s, const char* suf) {
  size_t n = strlen(suf);
  return s.size() >= n && memcmp(s.data() + s.size() - n, suf, n) == 0;
}

bool (*Output::big_text)(Output&, const char*, size_t) = nullptr;

void Output::open(const std::string& p, bool rewrite) {
  path = p;
  if (p == "-") {
    fp = stdout;
    return;
  }
  struct stat st;
  if (stat(p.c_str(), &st) == 0 && !rewrite)
    fail("File `" + p + "` already exists, please add `-r` to rewrite it."); /* errors.rs:25 */
  if (ends_with(p, ".bz2") || ends_with(p, ".xz"))
    fail("IO error:bz2 / xz output is not built into this engine (plain or .gz only)");
  bgzf = ends_with(p, ".gz");
  fp = fopen(p.c_str(), "wb");
  if (!fp) fail("IO error:cannot create `" + p + "`");
}

/* one BGZF member of n <= 0xff00 bytes by zlib (level 6: the reference's, utils.rs:192; stored blocks where deflate would not
 * fit the 64 KiB a member may have) */
static void bgzf_member_host(FILE* fp, const char* p, size_t n) {
  static const size_t kCap = 65536;
  unsigned char buf[kCap];
  size_t pay = 0;
  for (int level : {6, 0}) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) fail("IO error:zlib deflateInit2 failed");
    zs.next_in = (Bytef*)p;
    zs.avail_in = (uInt)n;
    zs.next_out = buf + 18;
    zs.avail_out = (uInt)(kCap - 18 - 8);
    const int r = deflate(&zs, Z_FINISH);
    pay = zs.total_out;
    deflateEnd(&zs);
    if (r == Z_STREAM_END) break;
    if (level == 0) fail("IO error:zlib deflate failed");
  }
  static const unsigned char head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
  memcpy(buf, head, 16);
  const uint32_t bsize = (uint32_t)(18 + pay + 8 - 1), crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)p, (uInt)n), isize = (uint32_t)n;
  buf[16] = (unsigned char)(bsize & 0xff);
  buf[17] = (unsigned char)(bsize >> 8);
  unsigned char* t = buf + 18 + pay;
  for (int k = 0; k < 4; k++) t[k] = (unsigned char)(crc >> (8 * k));
  for (int k = 0; k < 4; k++) t[4 + k] = (unsigned char)(isize >> (8 * k));
  if (fwrite(buf, 1, 18 + pay + 8, fp) != 18 + pay + 8) fail("IO error:write failed");
}
void Output::host_members(const char* p, size_t n) {
  static const size_t kIn = 0xff00;
  if (!pend.empty()) { /* top the waiting text up to a member first */
    const size_t take = std::min(n, kIn - pend.size());
    pend.append(p, take);
    p += take;
    n -= take;
    if (pend.size() < kIn) return;
    bgzf_member_host(fp, pend.data(), pend.size());
    pend.clear();
  }
  while (n >= kIn) {
    bgzf_member_host(fp, p, kIn);
    p += kIn;
    n -= kIn;
  }
  pend.append(p, n);
}
void Output::flush_pend() {
  if (pend.empty()) return;
  bgzf_member_host(fp, pend.data(), pend.size());
  pend.clear();
}
void Output::write(const char* p, size_t n) {
  if (!n) return;
  if (bgzf) {
    if (n >= kBigText && big_text && big_text(*this, p, n)) return;
    host_members(p, n);
  } else if (fwrite(p, 1, n, fp) != n) {
    fail("IO error:write failed");
  }
}
int Output::plain_fd(uint64_t* pos) {
  if (bgzf || !fp || fp == stdout) return -1;
  fflush(fp);
  const long at = ftell(fp);
  if (at < 0) return -1;
  *pos = (uint64_t)at;
  return fileno(fp);
}
int Output::raw_fd(uint64_t* pos) {
  if (!bgzf || !fp) return -1;
  flush_pend();
  fflush(fp);
  const long at = ftell(fp);
  if (at < 0) return -1;
  *pos = (uint64_t)at;
  return fileno(fp);
}
void Output::advance(uint64_t n) {
  if (fp && fp != stdout && fseek(fp, (long)n, SEEK_CUR) != 0) fail("IO error:seek failed");
}
void Output::close() {
  if (bgzf && fp) {
    flush_pend();
    static const unsigned char eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                          0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fwrite(eof, 1, sizeof eof, fp) != sizeof eof) fail("IO error:write failed");
  }
  if (fp && fp != stdout && fclose(fp) != 0) {
    fp = nullptr;
    fail("IO error:write failed");
  }
  if (fp == stdout) fflush(stdout);
  fp = nullptr;
}

/* ------------------------------------------------------------------------------------------ */
/* numbers                                                                                     */
/* ------------------------------------------------------------------------------------------ */
void append_u64(std::string& s, uint64_t v) {
  char b[24];
  auto r = std::to_chars(b, b + sizeof b, v);
  s.append(b, r.ptr);
}

/* Rust's u64::from_str: optional '+', at least one digit, no overflow */
static bool parse_u64(const std::string& s, uint64_t* out) {
  size_t i = 0;
  if (i < s.size() && s[i] == '+') i++;
  if (i >= s.size()) return false;
  uint64_t v = 0;
  for (; i < s.size(); i++) {
    if (s[i] < '0' || s[i] > '9') return false;
    uint64_t d = (uint64_t)(s[i] - '0');
    if (v > (UINT64_MAX - d) / 10) return false;
    v = v * 10 + d;
  }
  *out = v;
  return true;
}

/* ryu::Buffer::format(f32) -> pretty::format32 (ryu 1.0.14): shortest digits, then
 *   0 <= k && kk <= 13 : digits, zeros, ".0"        0 < kk <= 13 : dddd.ddd
 *   -6 < kk <= 0       : 0.000ddd                   else d[.ddd]e<exp>
 * (k = decimal exponent of the last digit, kk = position of the decimal point). */
std::string format_f32(float f) {
  if (std::isnan(f)) return "NaN";
  if (std::isinf(f)) return f < 0 ? "-inf" : "inf";
  if (f == 0.0f) return std::signbit(f) ? "-0.0" : "0.0";
  char b[64];
  auto r = std::to_chars(b, b + sizeof b, f, std::chars_format::scientific);
  std::string sci(b, r.ptr); /* [-]d[.ddd]e[+-]xx */
  std::string out;
  size_t p = 0;
  if (sci[0] == '-') {
    out.push_back('-');
    p = 1;
  }
  size_t epos = sci.find('e', p);
  std::string digits;
  for (size_t i = p; i < epos; i++)
    if (sci[i] != '.') digits.push_back(sci[i]);
  int exp10 = atoi(sci.c_str() + epos + 1);
  int len = (int)digits.size();
  int k = exp10 - (len - 1);
  int kk = len + k;
  if (0 <= k && kk <= 13) {
    out += digits;
    out.append((size_t)k, '0');
    out += ".0";
  } else if (0 < kk && kk <= 13) {
    out.append(digits, 0, (size_t)kk);
    out.push_back('.');
    out.append(digits, (size_t)kk, std::string::npos);
  } else if (-6 < kk && kk <= 0) {
    out += "0.";
    out.append((size_t)(-kk), '0');
    out += digits;
  } else if (len == 1) {
    out += digits;
    out.push_back('e');
    out += std::to_string(kk - 1);
  } else {
    out.push_back(digits[0]);
    out.push_back('.');
    out.append(digits, 1, std::string::npos);
    out.push_back('e');
    out += std::to_string(kk - 1);
  }
  return out;
}

/* ryu::Buffer::format(f64) -> pretty::format64: as format32 with the bounds 16 and -5 */
std::string format_f64(double f) {
  if (std::isnan(f)) return "NaN";
  if (std::isinf(f)) return f < 0 ? "-inf" : "inf";
  if (f == 0.0) return std::signbit(f) ? "-0.0" : "0.0";
  char b[64];
  auto r = std::to_chars(b, b + sizeof b, f, std::chars_format::scientific);
  std::string sci(b, r.ptr);
  std::string out;
  size_t p = 0;
  if (sci[0] == '-') {
    out.push_back('-');
    p = 1;
  }
  size_t epos = sci.find('e', p);
  std::string digits;
  for (size_t i = p; i < epos; i++)
    if (sci[i] != '.') digits.push_back(sci[i]);
  int exp10 = atoi(sci.c_str() + epos + 1);
  int len = (int)digits.size();
  int k = exp10 - (len - 1);
  int kk = len + k;
  if (0 <= k && kk <= 16) {
    out += digits;
    out.append((size_t)k, '0');
    out += ".0";
  } else if (0 < kk && kk <= 16) {
    out.append(digits, 0, (size_t)kk);
    out.push_back('.');
    out.append(digits, (size_t)kk, std::string::npos);
  } else if (-5 < kk && kk <= 0) {
    out += "0.";
    out.append((size_t)(-kk), '0');
    out += digits;
  } else if (len == 1) {
    out += digits;
    out.push_back('e');
    out += std::to_string(kk - 1);
  } else {
    out.push_back(digits[0]);
    out.push_back('.');
    out.append(digits, 1, std::string::npos);
    out.push_back('e');
    out += std::to_string(kk - 1);
  }
  return out;
}

void append_csv_field(std::string& s, const std::string& f, char delim) {
  bool need = false;
  for (char c : f)
    if (c == delim || c == '"' || c == '\n' || c == '\r') {
      need = true;
      break;
    }
  if (!need) {
    s += f;
    return;
  }
  s.push_back('"');
  for (char c : f) {
    if (c == '"') s.push_back('"');
    s.push_back(c);
  }
  s.push_back('"');
}

/* natord 1.0.9 compare(): whitespace skipped, digit runs compared numerically — left-aligned
 * when either run starts with '0', otherwise longest run wins, then first difference */
static bool is_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }
int natord_compare(const std::string& a, const std::string& b) {
  size_t i = 0, j = 0;
  auto digit = [](const std::string& s, size_t k) { return k < s.size() && s[k] >= '0' && s[k] <= '9'; };
  for (;;) {
    while (i < a.size() && is_ws((unsigned char)a[i])) i++;
    while (j < b.size() && is_ws((unsigned char)b[j])) j++;
    bool ea = i >= a.size(), eb = j >= b.size();
    if (ea && eb) return 0;
    if (ea) return -1;
    if (eb) return 1;
    unsigned char ca = (unsigned char)a[i++], cb = (unsigned char)b[j++];
    bool da = ca >= '0' && ca <= '9', db = cb >= '0' && cb <= '9';
    if (da && db) {
      if (ca == '0' || cb == '0') { /* compare_left */
        if (ca != cb) return ca < cb ? -1 : 1;
        for (;;) {
          bool la = digit(a, i), lb = digit(b, j);
          if (la && lb) {
            if (a[i] != b[j]) return a[i] < b[j] ? -1 : 1;
            i++;
            j++;
          } else if (la) {
            return 1;
          } else if (lb) {
            return -1;
          } else {
            break;
          }
        }
      } else { /* compare_right */
        int bias = ca == cb ? 0 : (ca < cb ? -1 : 1);
        for (;;) {
          bool la = digit(a, i), lb = digit(b, j);
          if (la && lb) {
            if (bias == 0 && a[i] != b[j]) bias = a[i] < b[j] ? -1 : 1;
            i++;
            j++;
          } else if (la) {
            return 1;
          } else if (lb) {
            return -1;
          } else {
            if (bias != 0) return bias;
            break;
          }
        }
      }
    } else if (ca != cb) {
      return ca < cb ? -1 : 1;
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* PAF (paf.rs)                                                                                */
/* ------------------------------------------------------------------------------------------ */
std::vector<PafRecord> parse_paf(const std::string& text) { return parse_paf(text, 0, 0, 0); }

std::vector<PafRecord> parse_paf(const std::string& text, uint64_t rec0, uint64_t line0, uint64_t byte0) {
  std::vector<PafRecord> out;
  size_t p = 0, n = text.size();
  uint64_t recno = rec0, line = line0 + 1;
  std::vector<std::string> fields;
  while (p < n) {
    /* record terminators: \n, \r\n, \r; blank lines are skipped */
    if (text[p] == '\n' || text[p] == '\r') {
      if (text[p] == '\n') line++;
      p++;
      continue;
    }
    if (text[p] == '#') { /* comment line */
      while (p < n && text[p] != '\n' && text[p] != '\r') p++;
      continue;
    }
    uint64_t rec_byte = byte0 + p;
    fields.clear();
    for (;;) {
      std::string f;
      if (p < n && text[p] == '"') { /* quoted field, "" = one quote */
        p++;
        while (p < n) {
          if (text[p] == '"') {
            if (p + 1 < n && text[p + 1] == '"') {
              f.push_back('"');
              p += 2;
            } else {
              p++;
              break;
            }
          } else {
            f.push_back(text[p++]);
          }
        }
        while (p < n && text[p] != '\t' && text[p] != '\n' && text[p] != '\r') f.push_back(text[p++]);
      } else {
        size_t s = p;
        while (p < n && text[p] != '\t' && text[p] != '\n' && text[p] != '\r') p++;
        f.assign(text, s, p - s);
      }
      fields.push_back(std::move(f));
      if (p < n && text[p] == '\t') {
        p++;
        continue;
      }
      break;
    }
    auto bad = [&](size_t field, const std::string& why) {
      fail("CSV deserialize error by: CSV deserialize error: record " + std::to_string(recno) +
           " (line: " + std::to_string(line) + ", byte: " + std::to_string(rec_byte) + "): field " +
           std::to_string(field) + ": " + why);
    };
    if (fields.size() < 12)
      fail("CSV deserialize error by: CSV deserialize error: record " + std::to_string(recno) +
           " (line: " + std::to_string(line) + ", byte: " + std::to_string(rec_byte) +
           "): invalid length " + std::to_string(fields.size()) + ", expected struct PafRecord with 13 elements");
    PafRecord r;
    auto num = [&](size_t k, uint64_t* v) {
      if (!parse_u64(fields[k], v)) bad(k, fields[k].empty() ? "cannot parse integer from empty string"
                                                             : "invalid digit found in string");
    };
    r.query_name = fields[0];
    num(1, &r.query_length);
    num(2, &r.query_start);
    num(3, &r.query_end);
    if (fields[4] == "+")
      r.neg = false;
    else if (fields[4] == "-")
      r.neg = true;
    else
      bad(4, "unknown variant `" + fields[4] + "`, expected `+` or `-`");
    r.target_name = fields[5];
    num(6, &r.target_length);
    num(7, &r.target_start);
    num(8, &r.target_end);
    num(9, &r.matches);
    num(10, &r.block_length);
    num(11, &r.mapq);
    r.tags.assign(fields.begin() + 12, fields.end());
    out.push_back(std::move(r));
    recno++;
  }
  return out;
}

/* paf.rs:159-218; the regex (:[0-9]+|\*[a-z][a-z]|[=\+\-][A-Za-z]+) as a leftmost scanner */
std::string cs_to_cigar(const std::string& cs) {
  std::string cigar;
  char last_op = 'M';
  uint64_t last_len = 0;
  auto flush = [&]() {
    if (last_len > 0) {
      append_u64(cigar, last_len);
      cigar.push_back(last_op);
    }
  };
  auto lower = [](char c) { return c >= 'a' && c <= 'z'; };
  auto alpha = [](char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); };
  auto dig = [](char c) { return c >= '0' && c <= '9'; };
  size_t i = 0, n = cs.size();
  while (i < n) {
    char c = cs[i];
    if (c == ':' && i + 1 < n && dig(cs[i + 1])) {
      size_t j = i + 1;
      uint64_t length = 0;
      while (j < n && dig(cs[j])) length = length * 10 + (uint64_t)(cs[j++] - '0');
      if (last_op == 'M') {
        last_len += length;
      } else {
        flush();
        last_op = 'M';
        last_len = length;
      }
      i = j;
    } else if (c == '*' && i + 2 < n && lower(cs[i + 1]) && lower(cs[i + 2])) {
      if (last_op == 'X') {
        last_len += 1;
      } else {
        flush();
        last_op = 'X';
        last_len = 1;
      }
      i += 3;
    } else if ((c == '=' || c == '+' || c == '-') && i + 1 < n && alpha(cs[i + 1])) {
      size_t j = i + 1;
      while (j < n && alpha(cs[j])) j++;
      if (c != '=') { /* '=' parts are matched but ignored, paf.rs:209 */
        flush();
        append_u64(cigar, (uint64_t)(j - i - 1));
        cigar.push_back(c == '-' ? 'D' : 'I');
        last_len = 0;
        last_op = 'M';
      }
      i = j;
    } else {
      i++;
    }
  }
  flush();
  return cigar;
}

std::string paf_cigar_string(const PafRecord& r, int* err) {
  *err = 0;
  for (const auto& t : r.tags)
    if (t.compare(0, 5, "cg:Z:") == 0) return t;
  for (const auto& t : r.tags)
    if (t.compare(0, 5, "cs:Z:") == 0) return "cg:Z:" + cs_to_cigar(t.substr(5));
  *err = 1;
  return "";
}

/* ------------------------------------------------------------------------------------------ */
/* MAF (maf.rs)                                                                                */
/* ------------------------------------------------------------------------------------------ */
static std::vector<std::string> split_ws(const std::string& line) {
  std::vector<std::string> f;
  size_t i = 0, n = line.size();
  while (i < n) {
    while (i < n && is_ws((unsigned char)line[i])) i++;
    size_t s = i;
    while (i < n && !is_ws((unsigned char)line[i])) i++;
    if (i > s) f.emplace_back(line, s, i - s);
  }
  return f;
}

static MafSLine parse_sline(const std::string& line) { /* maf.rs:138-211 */
  /* the reference takes the tokens left to right and fails at the FIRST one that is absent or malformed: a short
   * line with a bad number reports the number, not the missing field behind it */
  static const char* names[] = {"mode", "name", "start", "align_size", "strand", "size", "seq"};
  std::vector<std::string> f = split_ws(line);
  auto need = [&](size_t k) -> const std::string& {
    if (k >= f.size()) fail(std::string("Parse MAF error by: S-line Filed `") + names[k] + "` Missing");
    return f[k];
  };
  MafSLine s;
  (void)need(0);
  s.name = need(1);
  if (!parse_u64(need(2), &s.start)) fail("Parse `" + f[2] + "` Into Integer Error");
  if (!parse_u64(need(3), &s.align_size)) fail("Parse `" + f[3] + "` Into Integer Error");
  const std::string& st = need(4);
  if (st == "+")
    s.neg = false;
  else if (st == "-")
    s.neg = true;
  else
    fail("Parse Strand `" + st + "` Error");
  if (!parse_u64(need(5), &s.size)) fail("Parse `" + f[5] + "` Into Integer Error");
  s.seq = need(6);
  if (f.size() > 7) fail("Parse MAF error by: Surplus Filed > 7");
  return s;
}

std::vector<MafRecord> parse_maf(const std::string& text, std::string* header) {
  std::vector<MafRecord> out;
  size_t p = 0, n = text.size();
  auto next_line = [&](std::string* line) -> bool { /* BufRead::lines: strips \n and \r\n */
    if (p >= n) return false;
    size_t e = text.find('\n', p);
    size_t end = e == std::string::npos ? n : e;
    size_t le = end;
    if (le > p && text[le - 1] == '\r') le--;
    line->assign(text, p, le - p);
    p = e == std::string::npos ? n : e + 1;
    return true;
  };
  std::string line;
  if (next_line(&line) && header) *header = line; /* the first line is always the header */
  while (next_line(&line)) {
    if (line.empty() || line[0] != 's') continue;
    MafRecord rec;
    rec.slines.push_back(parse_sline(line));
    while (next_line(&line)) {
      if (!line.empty() && line[0] == 's')
        rec.slines.push_back(parse_sline(line));
      else
        break; /* the terminating line is consumed and dropped */
    }
    out.push_back(std::move(rec));
  }
  return out;
}

/* ------------------------------------------------------------------------------------------ */
/* chain (chain.rs)                                                                            */
/* ------------------------------------------------------------------------------------------ */
/* From<nom::Err<Error<&str>>> (errors.rs:88-96): the message quotes the first 10 bytes of the
 * unparsed input and the slice panics when fewer are left */
static std::string nom_error(const char* kind, const std::string& text, size_t at) {
  if (text.size() - at < 10)
    return "panic: byte index 10 is out of range of the unparsed input (errors.rs:92)";
  return std::string("Format error ") + kind + " at: " + text.substr(at, 10) + " Parse Error by rust::nom, please check";
}
/* f64::from_str accepts: [+-] (inf | infinity | nan | digits[.digits][e[+-]digits] | .digits[e..]) */
static bool valid_f64(const std::string& s) {
  size_t i = 0, n = s.size();
  if (i < n && (s[i] == '+' || s[i] == '-')) i++;
  std::string rest = s.substr(i);
  for (auto& c : rest) c = (char)tolower((unsigned char)c);
  if (rest == "inf" || rest == "infinity" || rest == "nan") return true;
  size_t d0 = i;
  while (i < n && isdigit((unsigned char)s[i])) i++;
  size_t nd = i - d0;
  if (i < n && s[i] == '.') {
    i++;
    size_t f0 = i;
    while (i < n && isdigit((unsigned char)s[i])) i++;
    nd += i - f0;
  }
  if (nd == 0) return false;
  if (i < n && (s[i] == 'e' || s[i] == 'E')) {
    i++;
    if (i < n && (s[i] == '+' || s[i] == '-')) i++;
    size_t e0 = i;
    while (i < n && isdigit((unsigned char)s[i])) i++;
    if (i == e0) return false;
  }
  return i == n;
}

std::vector<ChainRecord> parse_chain(const std::string& text, std::string* err) {
  std::vector<ChainRecord> out;
  err->clear();
  const size_t n = text.size();
  size_t p = 0;
  bool stop[256] = {false}; /* is_not("chain\n") */
  for (const char* c = "chain\n"; *c; c++) stop[(unsigned char)*c] = true;
  auto line_ending = [&](size_t at) -> size_t { /* length of "\n" / "\r\n" at `at`, 0 if none */
    if (at < n && text[at] == '\n') return 1;
    if (at + 1 < n && text[at] == '\r' && text[at + 1] == '\n') return 2;
    return 0;
  };
  while (p < n) {
    /* tag("chain") */
    if (text.compare(p, 5, "chain") != 0) {
      *err = nom_error("Tag", text, p);
      return out;
    }
    p += 5;
    /* not_line_ending: up to the first \r or \n; a lone \r is an error */
    size_t e = p;
    while (e < n && text[e] != '\n' && text[e] != '\r') e++;
    if (e < n && text[e] == '\r' && !(e + 1 < n && text[e + 1] == '\n')) {
      *err = nom_error("Tag", text, p);
      return out;
    }
    /* parse_header (chain.rs:206-322): whitespace-separated, surplus fields ignored */
    static const char* names[] = {"score", "target_name", "target_size", "target_strand", "target_start", "target_end",
                                  "query_name", "query_size", "query_strand", "query_start", "query_end", "chain_id"};
    std::vector<std::string> f = split_ws(text.substr(p, e - p));
    ChainRecord r;
    uint64_t* ints[12] = {nullptr, nullptr, &r.target_size, nullptr, &r.target_start, &r.target_end,
                          nullptr, &r.query_size, nullptr, &r.query_start, &r.query_end, &r.chain_id};
    for (size_t k = 0; k < 12; k++) {
      if (k >= f.size()) {
        *err = std::string("Parse Chain Error By: Chain Line Field `") + names[k] + "` Missing";
        return out;
      }
      if (k == 0) {
        if (!valid_f64(f[0])) {
          *err = "Parse `" + f[0] + "` Into Float Error";
          return out;
        }
      } else if (k == 1) {
        r.target_name = f[1];
      } else if (k == 6) {
        r.query_name = f[6];
      } else if (k == 3 || k == 8) {
        if (f[k] != "+" && f[k] != "-") {
          *err = "Parse Strand `" + f[k] + "` Error";
          return out;
        }
        (k == 3 ? r.target_neg : r.query_neg) = f[k] == "-";
      } else if (!parse_u64(f[k], ints[k])) {
        *err = "Parse `" + f[k] + "` Into Integer Error";
        return out;
      }
    }
    /* line_ending */
    size_t le = line_ending(e);
    if (!le) {
      *err = nom_error("CrLf", text, e);
      return out;
    }
    p = e + le;
    /* fold_many1(terminated(is_not("chain\n"), line_ending)) */
    size_t n_lines = 0;
    std::string line_err;
    for (;;) {
      size_t q = p;
      while (q < n && !stop[(unsigned char)text[q]]) q++;
      if (q == p) break;
      size_t l2 = line_ending(q);
      if (!l2) break;
      if (line_err.empty()) { /* after an error the fold keeps consuming lines but ignores them */
        uint64_t v[3] = {0, 0, 0};
        size_t a = p, nf = 0;
        for (; nf < 3; nf++) { /* split_whitespace: up to three tokens, the rest is ignored */
          while (a < q && is_ws((unsigned char)text[a])) a++;
          if (a >= q) break;
          size_t b = a;
          uint64_t x = 0;
          bool ok = true;
          size_t dstart = b < q && text[b] == '+' ? b + 1 : b;
          for (b = dstart; b < q && !is_ws((unsigned char)text[b]); b++) {
            const unsigned dgt = (unsigned)(text[b] - '0');
            if (dgt > 9 || x > (UINT64_MAX - dgt) / 10) ok = false;
            if (ok) x = x * 10 + dgt;
          }
          if (!ok || b == dstart) {
            line_err = "Parse `" + text.substr(a, b - a) + "` Into Integer Error";
            break;
          }
          v[nf] = x;
          a = b;
        }
        if (line_err.empty() && nf == 0) line_err = "Parse Chain Error By: Chain Line Field `size` Missing";
        if (line_err.empty()) r.lines.insert(r.lines.end(), v, v + 3);
      }
      n_lines++;
      p = q + l2;
    }
    if (n_lines == 0) {
      *err = nom_error("Many1", text, p);
      return out;
    }
    if (!line_err.empty()) {
      *err = line_err;
      return out;
    }
    /* take_while(|x| x != 'c') */
    while (p < n && text[p] != 'c') p++;
    out.push_back(std::move(r));
  }
  return out;
}

/* ------------------------------------------------------------------------------------------ */
/* FASTA                                                                                       */
/* ------------------------------------------------------------------------------------------ */
void Faidx::load(const std::string& path) {
  std::string text = read_all_parallel(path);
  pool.clear();
  pool.reserve(text.size());
  contigs.clear();
  size_t p = 0, n = text.size();
  std::string cur;
  uint64_t cur_off = 0;
  bool have = false;
  auto close = [&]() { /* a repeated name keeps its FIRST sequence, as htslib's fai_build does */
    if (have) contigs.emplace(cur, Contig{(uint64_t)pool.size() - cur_off, cur_off});
  };
  while (p < n) {
    size_t e = text.find('\n', p);
    size_t end = e == std::string::npos ? n : e;
    size_t le = end;
    if (le > p && text[le - 1] == '\r') le--;
    if (le > p && text[p] == '>') {
      close();
      size_t s = p + 1, q = s;
      while (q < le && !is_ws((unsigned char)text[q])) q++;
      cur.assign(text, s, q - s);
      cur_off = pool.size();
      have = true;
    } else if (have) {
      pool.append(text, p, le - p);
    }
    p = e == std::string::npos ? n : e + 1;
  }
  close();
}

void Faidx::set_table(const std::string& text, const uint64_t* tab, size_t n) {
  contigs.clear();
  for (size_t k = 0; k < n; k++) {
    const uint64_t hs = tab[4 * k], he = tab[4 * k + 1];
    size_t s = (size_t)hs + 1, q = s;
    size_t le = (size_t)he;
    if (le > s && text[le - 1] == '\r') le--;
    while (q < le && !is_ws((unsigned char)text[q])) q++;
    /* a repeated name keeps its FIRST sequence, as htslib's fai_build does */
    contigs.emplace(std::string(text, s, q - s), Contig{tab[4 * k + 3], tab[4 * k + 2]});
  }
}

void Faidx::fetch(const std::string& name, uint64_t beg_u, uint64_t end_u, uint64_t* off,
                  uint64_t* len) const {
  if (beg_u > (uint64_t)INT64_MAX || end_u > (uint64_t)INT64_MAX)
    fail("HTS library error by The given position is too large to be converted to i64");
  auto it = contigs.find(name);
  if (it == contigs.end()) fail("HTS library error by sequence `" + name + "` not found in the FASTA index");
  int64_t L = (int64_t)it->second.len, beg = (int64_t)beg_u, end = (int64_t)end_u;
  /* faidx_adjust_position (htslib), end inclusive */
  if (end < beg) beg = end;
  if (beg < 0)
    beg = 0;
  else if (L <= beg)
    beg = L;
  if (end < 0)
    end = 0;
  else if (L <= end)
    end = L - 1;
  int64_t n = end + 1 - beg;
  if (n < 0) n = 0;
  *off = it->second.pool_off + (uint64_t)beg;
  *len = (uint64_t)n;
}

/* ------------------------------------------------------------------------------------------ */
/* MAF index (JSON) -> reference contigs                                                        */
/* ------------------------------------------------------------------------------------------ */
namespace {
struct Json { /* just enough of a JSON reader for tools/index.rs:78-95 */
  const std::string& t;
  size_t p = 0;
  explicit Json(const std::string& text) : t(text) {}
  void ws() {
    while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\r' || t[p] == '\t')) p++;
  }
  [[noreturn]] void bad() { fail("maf index: invalid JSON at byte " + std::to_string(p)); }
  bool eat(char c) {
    ws();
    if (p < t.size() && t[p] == c) {
      p++;
      return true;
    }
    return false;
  }
  void need(char c) {
    if (!eat(c)) bad();
  }
  std::string str() {
    need('"');
    std::string o;
    while (p < t.size() && t[p] != '"') {
      if (t[p] == '\\') {
        if (++p >= t.size()) bad();
        switch (t[p]) {
          case 'n': o.push_back('\n'); break;
          case 't': o.push_back('\t'); break;
          case 'r': o.push_back('\r'); break;
          case 'b': o.push_back('\b'); break;
          case 'f': o.push_back('\f'); break;
          case 'u': { /* \uXXXX -> UTF-8 (BMP only; names are ASCII in practice) */
            if (p + 4 >= t.size()) bad();
            unsigned cp = (unsigned)strtoul(t.substr(p + 1, 4).c_str(), nullptr, 16);
            p += 4;
            if (cp < 0x80)
              o.push_back((char)cp);
            else if (cp < 0x800) {
              o.push_back((char)(0xC0 | (cp >> 6)));
              o.push_back((char)(0x80 | (cp & 63)));
            } else {
              o.push_back((char)(0xE0 | (cp >> 12)));
              o.push_back((char)(0x80 | ((cp >> 6) & 63)));
              o.push_back((char)(0x80 | (cp & 63)));
            }
            break;
          }
          default: o.push_back(t[p]);
        }
        p++;
      } else {
        o.push_back(t[p++]);
      }
    }
    if (p >= t.size()) bad();
    p++;
    return o;
  }
  /* skips any value; numbers / literals are returned as text */
  std::string skip() {
    ws();
    if (p >= t.size()) bad();
    if (t[p] == '"') return str();
    if (t[p] == '{') {
      p++;
      if (eat('}')) return "";
      do {
        str();
        need(':');
        skip();
      } while (eat(','));
      need('}');
      return "";
    }
    if (t[p] == '[') {
      p++;
      if (eat(']')) return "";
      do skip();
      while (eat(','));
      need(']');
      return "";
    }
    size_t s = p;
    while (p < t.size() && t[p] != ',' && t[p] != '}' && t[p] != ']' && t[p] != ' ' && t[p] != '\n' && t[p] != '\r' &&
           t[p] != '\t')
      p++;
    if (p == s) bad();
    return t.substr(s, p - s);
  }
};
}  // namespace

std::vector<std::pair<std::string, uint64_t>> maf_index_ref_contigs(const std::string& path) {
  std::vector<std::pair<std::string, uint64_t>> out;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return out;
  fclose(f);
  std::string text = read_all(&path);
  Json j(text);
  j.need('{');
  if (!j.eat('}')) {
    do {
      std::string name = j.str();
      j.need(':');
      j.need('{');
      uint64_t size = 0;
      bool isref = false;
      if (!j.eat('}')) {
        do {
          std::string key = j.str();
          j.need(':');
          std::string v = j.skip();
          if (key == "size") size = strtoull(v.c_str(), nullptr, 10);
          if (key == "isref") isref = v == "true";
        } while (j.eat(','));
        j.need('}');
      }
      if (isref) out.emplace_back(name, size);
    } while (j.eat(','));
    j.need('}');
  }
  std::stable_sort(out.begin(), out.end(),
                   [](const auto& a, const auto& b) { return natord_compare(a.first, b.first) < 0; });
  return out;
}

/* ------------------------------------------------------------------------------------------ */
/* stat (common.rs:116-140, stat.rs)                                                           */
/* ------------------------------------------------------------------------------------------ */
RecStat recstat_from(const wga_cigar_counts& c) {
  RecStat r;
  r.matched = c.match;
  r.mismatched = c.mismatch;
  r.ins_event = c.ins_ev;
  r.del_event = c.del_ev;
  r.ins_size = c.ins_bp;
  r.del_size = c.del_bp;
  r.inv_ins_event = c.inv_ins_ev;
  r.inv_ins_size = c.inv_ins_bp;
  r.inv_del_event = c.inv_del_ev;
  r.inv_del_size = c.inv_del_bp;
  r.aligned_size = r.matched + r.mismatched + r.del_size + r.inv_del_size;
  uint64_t query_align = r.matched + r.mismatched + r.ins_size + r.inv_ins_size;
  r.inv_event = c.inv_ev;
  if (r.inv_event != 0) r.inv_size = (float)(r.aligned_size + query_align) / (float)(r.inv_event + 1);
  return r;
}

namespace {
struct Statistic { /* stat.rs:27-50, field order = column order */
  std::string ref_name;
  uint64_t ref_size = 0, ref_start = 0;
  std::string query_name;
  uint64_t query_size = 0, query_start = 0, aligned_size = 0, unaligned_size = 0;
  float identity = 0, similarity = 0;
  uint64_t matched = 0, mismatched = 0, ins_event = 0, del_event = 0, ins_size = 0, del_size = 0,
           inv_event = 0;
  float inv_size = 0;
  uint64_t inv_ins_event = 0, inv_ins_size = 0, inv_del_event = 0, inv_del_size = 0;
};
void add(Statistic& s, const RecStat& r) {
  s.aligned_size += r.aligned_size;
  s.matched += r.matched;
  s.mismatched += r.mismatched;
  s.ins_event += r.ins_event;
  s.del_event += r.del_event;
  s.ins_size += r.ins_size;
  s.del_size += r.del_size;
  s.inv_ins_event += r.inv_ins_event;
  s.inv_ins_size += r.inv_ins_size;
  s.inv_del_event += r.inv_del_event;
  s.inv_del_size += r.inv_del_size;
  s.inv_event += r.inv_event;
  s.inv_size += r.inv_size;
}
}  // namespace

std::string stat_tsv(const std::vector<StatInput>& recs, bool each) {
  std::vector<Statistic> rows;
  if (each) { /* split_final, stat.rs:129-164: unaligned_size stays 0 */
    for (const auto& in : recs) {
      Statistic s;
      s.ref_name = in.ref_name;
      s.ref_size = in.ref_size;
      s.ref_start = in.ref_start;
      s.query_name = in.query_name;
      s.query_size = in.query_size;
      s.query_start = in.query_start;
      add(s, in.rs);
      s.identity = (float)s.matched / (float)s.aligned_size;
      s.similarity = (float)(s.matched + s.mismatched) / (float)s.aligned_size;
      rows.push_back(std::move(s));
    }
  } else { /* merge_final_from_pair, stat.rs:167-223 (groups in first-appearance order; the
              reference iterates a HashMap, i.e. in random order) */
    std::map<std::tuple<std::string, uint64_t, std::string, uint64_t>, size_t> index;
    for (const auto& in : recs) {
      auto key = std::make_tuple(in.ref_name, in.ref_size, in.query_name, in.query_size);
      auto it = index.find(key);
      if (it == index.end()) {
        Statistic s;
        s.ref_name = in.ref_name;
        s.ref_size = in.ref_size;
        s.ref_start = in.ref_size; /* minima start from the sizes, stat.rs:186,189 */
        s.query_name = in.query_name;
        s.query_size = in.query_size;
        s.query_start = in.query_size;
        it = index.emplace(key, rows.size()).first;
        rows.push_back(std::move(s));
      }
      Statistic& s = rows[it->second];
      add(s, in.rs);
      if (in.ref_start < s.ref_start) s.ref_start = in.ref_start;
      if (in.query_start < s.query_start) s.query_start = in.query_start;
    }
    for (auto& s : rows) {
      s.unaligned_size = s.ref_size - s.aligned_size; /* wraps like release-mode Rust */
      s.identity = (float)s.matched / (float)s.aligned_size;
      s.similarity = (float)(s.matched + s.mismatched) / (float)s.aligned_size;
    }
  }
  std::stable_sort(rows.begin(), rows.end(), [](const Statistic& a, const Statistic& b) {
    return natord_compare(a.ref_name, b.ref_name) < 0;
  });
  std::string out;
  if (rows.empty()) return out; /* csv writes the header with the first row only */
  out += "ref_name\tref_size\tref_start\tquery_name\tquery_size\tquery_start\taligned_size\t"
         "unaligned_size\tidentity\tsimilarity\tmatched\tmismatched\tins_event\tdel_event\tins_size\t"
         "del_size\tinv_event\tinv_size\tinv_ins_event\tinv_ins_size\tinv_del_event\tinv_del_size\n";
  for (const auto& s : rows) {
    append_csv_field(out, s.ref_name, '\t');
    out.push_back('\t');
    append_u64(out, s.ref_size);
    out.push_back('\t');
    append_u64(out, s.ref_start);
    out.push_back('\t');
    append_csv_field(out, s.query_name, '\t');
    out.push_back('\t');
    uint64_t a[] = {s.query_size, s.query_start, s.aligned_size, s.unaligned_size};
    for (uint64_t v : a) {
      append_u64(out, v);
      out.push_back('\t');
    }
    out += format_f32(s.identity);
    out.push_back('\t');
    out += format_f32(s.similarity);
    out.push_back('\t');
    uint64_t b[] = {s.matched, s.mismatched, s.ins_event, s.del_event, s.ins_size, s.del_size, s.inv_event};
    for (uint64_t v : b) {
      append_u64(out, v);
      out.push_back('\t');
    }
    out += format_f32(s.inv_size);
    out.push_back('\t');
    uint64_t c[] = {s.inv_ins_event, s.inv_ins_size, s.inv_del_event, s.inv_del_size};
    for (int k = 0; k < 4; k++) {
      append_u64(out, c[k]);
      out.push_back(k == 3 ? '\n' : '\t');
    }
  }
  return out;
}

/* ------------------------------------------------------------------------------------------ */
/* CIGAR error messages (errors.rs:51-60)                                                      */
/* ------------------------------------------------------------------------------------------ */
std::string cigar_error_message(int32_t err, const std::string& cigar, size_t tok_off, size_t tok_len) {
  std::string tok = cigar.substr(tok_off, tok_len);
  switch (err) {
    case WGA_REC_CIGAR_OP_INVALID: return "CIGAR OP `" + tok + "` invalid";
    case WGA_REC_PARSE_INT: return "Parse `" + tok + "` Into Integer Error";
    case WGA_REC_CIGAR_TAG_NOT_FOUND: return "CIGAR start tag not found";
    case WGA_REC_PANIC: return "panic: empty CIGAR (the reference panics in errors.rs:92)";
    default: return "CIGAR error";
  }
}

std::string cigar_op_token_at(const std::string& cigar, uint64_t op_idx) {
  /* packed ops of one text op can be several (split lengths): walk like the packer does */
  size_t p = 0, n = cigar.size();
  uint64_t k = 0;
  while (p < n) {
    size_t ls = p;
    while (p < n && cigar[p] >= '0' && cigar[p] <= '9') p++;
    uint64_t v = 0;
    for (size_t i = ls; i < p; i++) v = v * 10 + (uint64_t)(cigar[i] - '0');
    size_t os = p;
    while (p < n && !(cigar[p] >= '0' && cigar[p] <= '9')) p++;
    uint64_t pieces = v == 0 ? 1 : (v + WGA_OP_MAX_LEN - 1) / WGA_OP_MAX_LEN;
    if (op_idx < k + pieces) return cigar.substr(os, p - os);
    k += pieces;
  }
  return "";
}

}  // namespace wga
