/*
 * wga_host.hpp — host side of the drop-in: record types, text parsers and writers with the
 * reference's semantics (C++ because the image has no Rust toolchain).  No compute here: the
 * drivers in wgatools_main.cpp hand batches to libwgahip.so through include/wga_hip.h.
 *
 * Mirrors (reference file:line)
 *   PafRecord / PAFReader          parser/paf.rs:13-78      (csv crate: tab, flexible, '#' comments)
 *   get_cigar_string / cs_to_cigar parser/paf.rs:122-218
 *   MAFSLine / MAFRecord / reader  parser/maf.rs:65-73,138-211,216-220,371-421,424-478
 *   MAFWriter                      parser/maf.rs:543-582
 *   ChainRecord / chain_parser     parser/chain.rs:49-91,206-383 (nom 7 combinators)
 *   faidx fetch_seq_string         rust-htslib 0.44.1 faidx (htslib faidx_fetch_seq64)  [unpinned]
 *   RecStat::from                  parser/common.rs:116-140
 *   Statistic / merge / split      tools/stat.rs:27-50,129-223
 *   csv writer, ryu f32, natord    csv 1.2.2 / ryu 1.0.14 / natord 1.0.9                [unpinned]
 */
#ifndef WGA_HOST_HPP
#define WGA_HOST_HPP

#include <stdint.h>

#include <map>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/wga_hip.h"

namespace wga {

/* errors: message text of WGAError (errors.rs:8-86); thrown, caught in main, printed as
 * "<timestamp> ERROR <msg>" on stderr, exit code 1 (main.rs:14-21, log.rs:9-31) */
struct Error {
  std::string msg;
};
[[noreturn]] void fail(const std::string& msg);

/* ---- input / output (utils.rs:135-246: plain or gzip; "-" = stdin/stdout; -r to overwrite) --- */
std::string read_all(const std::string* path); /* nullptr = stdin */
/* the same bytes for a file; a BGZF file (`all.fa.gz` next to its .gzi / .fai, pseudomaf.rs:222) is inflated block by
 * block on all host cores — its blocks are independent gzip members */
std::string read_all_parallel(const std::string& path);
/* a BGZF file as it is, with its members' table (deflate stream offset / length, ISIZE, output offset per member) for the
 * device inflater (wga_bgzf_inflate): false when the file is not BGZF from end to end (then nothing is returned) */
struct BgzfMember {
  uint64_t in_off;
  uint32_t in_len, out_len;
  uint64_t out_off;
};
bool read_bgzf_image(const std::string& path, std::string& img, std::vector<BgzfMember>& members, uint64_t* total);
/* the member table alone (the file is walked header by header, nothing of it is kept): for readers that take the compressed
 * bytes a run of members at a time */
bool scan_bgzf_members(const std::string& path, std::vector<BgzfMember>& members, uint64_t* total, uint64_t* file_bytes);
uint32_t gzip_crc32(const void* p, size_t n); /* the CRC-32 of a gzip member's trailer (zlib's) */
struct Output {
  std::string path;
  /* `.gz` (utils.rs:201-209: the reference wraps the file in a gzip encoder): the file is a sequence of BGZF members — complete
   * gzip members of at most 64 KiB with the `BC` field — closed by BGZF's empty member.  Text that is already in HBM comes
   * compressed from the device (K18, wga_bgzf_compress) and is appended as it is (raw_fd); host text is collected in `pend`
   * and leaves as zlib level-6 members; host text of megabytes goes through the device as well once a command has one
   * (big_text, set by Dev::init).  Any gzip reader inflates the file to the bytes the plain output would hold. */
  bool bgzf = false;
  std::string pend;
  FILE* fp = nullptr;
  void open(const std::string& path, bool rewrite);
  void write(const char* p, size_t n);
  void write(const std::string& s) { write(s.data(), s.size()); }
  void close();
  /* a plain file (not stdout, not .gz): positional writes from several threads are possible.  Returns the file
   * descriptor after flushing what stdio holds and the offset where the next byte belongs, or -1. */
  int plain_fd(uint64_t* pos);
  /* the same for finished BGZF members going into a `.gz` file (host text waiting in `pend` leaves first); -1 for anything else */
  int raw_fd(uint64_t* pos);
  void advance(uint64_t n); /* the caller wrote n bytes at the position plain_fd / raw_fd reported */
  /* (out, text, bytes) -> true when the text went out through the device's deflate */
  static bool (*big_text)(Output&, const char*, size_t);
  static const size_t kBigText = (size_t)1 << 20; /* zlib takes 10-20 ms for this much; the trip through the device about one */

 private:
  void host_members(const char* p, size_t n); /* whole members of 0xff00 bytes, the rest into pend */
  void flush_pend();
};

/* ---- PAF ------------------------------------------------------------------------------------ */
struct PafRecord { /* paf.rs:50-65 */
  std::string query_name;
  uint64_t query_length = 0, query_start = 0, query_end = 0;
  bool neg = false;
  std::string target_name;
  uint64_t target_length = 0, target_start = 0, target_end = 0;
  uint64_t matches = 0, block_length = 0, mapq = 0;
  std::vector<std::string> tags;
};
/* csv-crate reading: records end at \n, \r\n or \r; empty lines and lines starting with '#' are
 * skipped; fields split on tabs with '"' quoting. */
std::vector<PafRecord> parse_paf(const std::string& text);
/* the same on a piece of a file: rec0 / line0 / byte0 = records, lines and bytes in front of the piece (they only
 * show in the csv crate's error text) */
std::vector<PafRecord> parse_paf(const std::string& text, uint64_t rec0, uint64_t line0, uint64_t byte0);

/* Reads a (plain or gzip) file or stdin in pieces that end at a line end: next() returns false at the end of the
 * input.  A piece holds at least `target` bytes unless the input ends; a piece that contains a '"' takes the rest of
 * the input with it (a quoted csv field may hold line ends). */
struct LineChunkReader {
  void* gz = nullptr; /* gzFile of a gzip input */
  int fd = -1;        /* a plain file is read with read(2): zlib's transparent mode copies at ~1.2 GB/s */
  uint64_t file_left = 0; /* ... and its size is known: a piece's buffer is reserved once instead of grown by doubling */
  bool is_stdin = false, eof = false;
  std::string carry;
  uint64_t bytes_before = 0, lines_before = 0; /* of the piece next() returned last */
  uint64_t next_bytes = 0, next_lines = 0;
  /* a producer of the input's bytes in the place of the file (a bgzipped input inflated on the device): returns the number
   * of bytes written to dst (at most want), 0 at the end */
  std::function<size_t(char* dst, size_t want)> source;
  void open(const std::string* path);
  bool next(std::string& piece, size_t target, size_t keep = 0); /* piece[0, keep) = the caller's prefix, kept */
  /* a piece the caller is done with: its memory serves a later piece (a fresh 256 MB buffer costs ~0.13 s of first-touch page
   * faults and zero fill, more than reading the file into it).  Any thread. */
  void recycle(std::string&& piece);
  ~LineChunkReader();
 private:
  std::mutex spare_mx;
  std::vector<std::string> spares;
};
/* paf.rs:122-141: the cg:Z: tag, or the cs:Z: tag converted; "" + err=1 when neither exists */
std::string paf_cigar_string(const PafRecord& r, int* err);
std::string cs_to_cigar(const std::string& cs);

/* ---- MAF ------------------------------------------------------------------------------------ */
struct MafSLine { /* maf.rs:65-73 */
  std::string name;
  uint64_t start = 0, align_size = 0;
  bool neg = false;
  uint64_t size = 0;
  std::string seq;              /* the row text (host reader) ... */
  const char* file = nullptr;   /* ... or a span of the input file, which stays in memory (device splitter) */
  uint64_t seq_off = 0, seq_len = 0;
  size_t seq_size() const { return file ? (size_t)seq_len : seq.size(); }
  const char* seq_data() const { return file ? file + seq_off : seq.data(); }
};
struct MafRecord { /* maf.rs:216-220 */
  uint64_t score = 255;
  std::vector<MafSLine> slines;
  size_t query_idx = 1;
  /* accessors of maf.rs:424-478 */
  const MafSLine& t() const { return slines[0]; }
  const MafSLine& q() const { return slines[query_idx]; }
  uint64_t query_start() const { return q().neg ? q().size - q().start - q().align_size : q().start; }
  uint64_t query_end() const { return q().neg ? q().size - q().start : q().start + q().align_size; }
};
/* maf.rs:25-36 + 371-421: first line is always the header; a block = maximal run of 's' lines */
std::vector<MafRecord> parse_maf(const std::string& text, std::string* header);

/* `<maf>.index` (tools/index.rs:78-95, serde_json map name -> {ivls,size,isref}): the (name, size) of
 * the entries with isref, natord-sorted (caller.rs:340-357).  Missing file -> empty. */
std::vector<std::pair<std::string, uint64_t>> maf_index_ref_contigs(const std::string& path);

/* ---- chain ---------------------------------------------------------------------------------- */
struct ChainRecord { /* chain.rs:49-55,76-91: header fields + data lines */
  std::string target_name, query_name;
  uint64_t target_size = 0, target_start = 0, target_end = 0;
  uint64_t query_size = 0, query_start = 0, query_end = 0;
  bool target_neg = false, query_neg = false;
  uint64_t chain_id = 0;
  std::vector<uint64_t> lines; /* 3 per data line: size, 2nd column (query_diff), 3rd column (target_diff) */
};
/* ChainRecords::next + chain_parser (chain.rs:58-73,206-383) with nom's behaviour: records start at
 * "chain", a data line is consumed only when it ends in a newline and holds none of the letters of
 * "chain", everything up to the next 'c' is skipped after the data lines.  Parsing stops at the first
 * error: *err receives the reference's message and the records before it are returned. */
std::vector<ChainRecord> parse_chain(const std::string& text, std::string* err);

/* ---- FASTA index ------------------------------------------------------------------------------ */
struct Faidx {
  struct Contig {
    uint64_t len, pool_off;
  };
  std::unordered_map<std::string, Contig> contigs;
  std::string pool; /* all contigs, newlines stripped, case preserved (host reader only) */
  void load(const std::string& path);
  /* device reader: the pool is built in HBM by wga_fasta_pool from the uploaded text; the table comes back as
   * (header start, header end, pool offset, length) per contig and only the names are read on the host */
  void set_table(const std::string& text, const uint64_t* tab /* n x 4 */, size_t n);
  /* faidx_fetch_seq64(name, beg, end inclusive) clipping: returns (pool offset, length) */
  void fetch(const std::string& name, uint64_t beg, uint64_t end_incl, uint64_t* off, uint64_t* len) const;
};

/* ---- formatting ------------------------------------------------------------------------------- */
void append_u64(std::string& s, uint64_t v);
/* ryu::Buffer::format(f32) as used by csv's serializer */
std::string format_f32(float f);
std::string format_f64(double f);
/* csv writer, QuoteStyle::Necessary */
void append_csv_field(std::string& s, const std::string& f, char delim);
/* natord::compare (natural order, digit runs numeric) */
int natord_compare(const std::string& a, const std::string& b);

/* ---- stat ------------------------------------------------------------------------------------- */
struct RecStat { /* common.rs:99-113 */
  uint64_t aligned_size = 0, matched = 0, mismatched = 0, ins_event = 0, del_event = 0,
           ins_size = 0, del_size = 0, inv_ins_event = 0, inv_ins_size = 0, inv_del_event = 0,
           inv_del_size = 0, inv_event = 0;
  float inv_size = 0.0f;
};
RecStat recstat_from(const wga_cigar_counts& c); /* common.rs:116-140 */
struct StatInput {
  std::string ref_name, query_name;
  uint64_t ref_size, query_size, ref_start, query_start;
  RecStat rs;
};
/* stat.rs:107-223: merge by (ref,ref_size,query,query_size) or one row per record, stable natord
 * sort by ref_name, TSV with header row (only when there is at least one row). */
std::string stat_tsv(const std::vector<StatInput>& recs, bool each);

/* error message for a tokeniser failure (errors.rs:51-60), given the CIGAR text after the tag */
std::string cigar_error_message(int32_t err, const std::string& cigar, size_t tok_off, size_t tok_len);
/* the op char of packed op index `idx` of a CIGAR text (for CigarOpInvalid raised by a kernel) */
std::string cigar_op_token_at(const std::string& cigar, uint64_t op_idx);

}  // namespace wga
#endif
