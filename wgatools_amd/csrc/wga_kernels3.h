/*
 * wga_kernels3.h — K16: the VCF rows of `call -f paf` (the caller behind the hot path).
 *
 * The reference turns every event of call_within_var_paf into one noodles-vcf record and prints it
 * (caller.rs:640-658 the <INV> row of a '-' record, :688-717 one row per column of an X op, :719-813 the
 * INS / DEL rows of indels longer than `svlen`; the text layout is noodles-vcf 0.43's, README.md:323-343):
 *
 *   <target>\t<pos>\t.\t<REF>\t<ALT>\t.\t.\t<INFO or .>\tGT:QI\t1|1:<query>@<a>[@<b>]@<P|N>\n
 *
 * K7 (wga_paf_call_events) leaves the events; this kernel formats them where they lie.  One wave per record,
 * one item per lane and step: item 0 is the record's <INV> row (nothing for a '+' record), item 1 + e its event e.
 * A step counts its items' bytes (digit counts, name lengths; the REF / ALT bases are read and checked), scans
 * them over the wave and — in the fill pass — every lane writes its rows into the wave's LDS buffer at the
 * place the scan gives; the buffer mirrors the output's position inside its 16-byte group and goes out with
 * 16-byte stores (a ragged head and tail by bytes).  A step whose text does not fit the buffer (long INS / DEL
 * sequences) is written straight to the output.
 *
 * Errors are the reference's: a REF / ALT slice outside the fetched sequences is its slice panic
 * (caller.rs:695-696,753-754,800-801), a base outside ACGTN (any case) is noodles-vcf's parse error; the first
 * item of a record that has one is reported (wga_vcf_err) and the record ends there.
 */
#ifndef WGA_KERNELS3_H
#define WGA_KERNELS3_H

#include "wga_k11_bridges.h"
#include "wga_k10_chain.h"

struct wga_vcf_rec_dev { /* = wga_vcf_rec (wga_hip.h) */
  u64 t_name_off, q_name_off; /* into `names` */
  u32 t_name_len, q_name_len;
  u64 t_start, t_end, q_start, q_end; /* the PAF columns */
  u64 t_off, t_len, q_off, q_len;     /* the fetched sequences inside the pools (paf.rs:221-237) */
};
struct wga_vcf_err_dev { /* = wga_vcf_err */
  u64 item;  /* WGA_NONE: clean */
  u32 kind;  /* 1 slice out of range, 2 base outside ACGTN */
  u32 ch;    /* kind 2: the byte as it stands in the sequence */
};

#define WGA_VCF_TB 8192u /* LDS text bytes per wave */

struct VcfCount {
  u64 n;
  __device__ __forceinline__ void c(u8) { n++; }
  __device__ __forceinline__ void dec(u64 v) { n += dec_digits(v); }
  __device__ __forceinline__ void str(const u8*, u32 len) { n += len; }
  /* bases are checked where they are counted; the first bad one is kept */
  __device__ __forceinline__ void bases(const u8* p, u64 len, u32* kind, u32* ch) {
    for (u64 k = 0; k < len; k++) {
      const u8 b = p[k], u = (b >= 'a' && b <= 'z') ? (u8)(b - 32) : b;
      if (u != 'A' && u != 'C' && u != 'G' && u != 'T' && u != 'N' && *kind == 0u) *kind = 2u, *ch = b;
    }
    n += len;
  }
};
struct VcfEmit {
  u8* p;
  __device__ __forceinline__ void c(u8 ch) { *p++ = ch; }
  __device__ __forceinline__ void dec(u64 v) {
    const u32 nd = dec_digits(v);
    dec_write(p, v, nd);
    p += nd;
  }
  __device__ __forceinline__ void str(const u8* s, u32 len) {
    for (u32 k = 0; k < len; k++) p[k] = s[k];
    p += len;
  }
  __device__ __forceinline__ void bases(const u8* s, u64 len, u32*, u32*) {
    for (u64 k = 0; k < len; k++) {
      const u8 b = s[k];
      p[k] = (b >= 'a' && b <= 'z') ? (u8)(b - 32) : b;
    }
    p += len;
  }
};
template <class S, u32 N>
__device__ __forceinline__ void vcf_lit(S& s, const char (&t)[N]) {
#pragma unroll
  for (u32 k = 0; k + 1u < N; k++) s.c((u8)t[k]);
}

struct VcfRecCtx {
  const u8 *t_name, *q_name, *ts, *qs;
  u32 t_name_len, q_name_len;
  u64 t_start, t_end, q_start, q_end, tn, qn, svlen, nops;
  const u32* rops;
  const u64* ev; /* the record's events */
  bool neg;
};

/* "<chro>\t<pos>\t.\t" */
template <class S>
__device__ __forceinline__ void vcf_row_open(S& s, const VcfRecCtx& r, u64 pos) {
  s.str(r.t_name, r.t_name_len);
  s.c((u8)'\t');
  s.dec(pos);
  vcf_lit(s, "\t.\t");
}
/* "\tGT:QI\t1|1:<query>@<a>[@<b>]@<P|N>\n" */
template <class S>
__device__ __forceinline__ void vcf_row_close(S& s, const VcfRecCtx& r, u64 a, u64 b, bool three) {
  vcf_lit(s, "\tGT:QI\t1|1:");
  s.str(r.q_name, r.q_name_len);
  s.c((u8)'@');
  s.dec(a);
  if (!three) {
    s.c((u8)'@');
    s.dec(b);
  }
  s.c((u8)'@');
  s.c(r.neg ? (u8)'N' : (u8)'P');
  s.c((u8)'\n');
}

/* the rows of one item; *kind != 0 on return: the item has the record's error (nothing more is written for it) */
template <class S>
__device__ __forceinline__ void vcf_item(S& s, const VcfRecCtx& r, u64 item, u32* kind, u32* ch) {
  if (item == 0) { /* caller.rs:640-658 */
    if (!r.neg) return;
    vcf_row_open(s, r, r.t_start + 1u);
    s.bases(r.ts, 1, kind, ch);
    vcf_lit(s, "\t<INV>\t.\t.\tSVTYPE=INV;END=");
    s.dec(r.t_end);
    vcf_row_close(s, r, r.q_start, r.q_end, false);
    return;
  }
  const u64* e = r.ev + 3u * (item - 1u);
  const u64 oi = e[0], tb = e[1], qb = e[2];
  const u32 code = r.rops[oi] & 15u;
  u64 len = r.rops[oi] >> 4;
  for (u64 j = oi + 1; j < r.nops && ((r.rops[j] & 15u) == WGA_OP_I_CONT || (r.rops[j] & 15u) == WGA_OP_D_CONT); j++)
    len += r.rops[j] >> 4;
  const u64 t_pos = r.t_start + tb, q_pos = r.q_start + qb;
  if (code == WGA_OP_X) { /* :688-717, one row per column */
    for (u64 x = 0; x < len; x++) {
      if (tb + x + 1u > r.tn || qb + x + 1u > r.qn) {
        if (*kind == 0u) *kind = 1u;
        return;
      }
      vcf_row_open(s, r, t_pos + x + 1u);
      s.bases(r.ts + tb + x, 1, kind, ch);
      s.c((u8)'\t');
      s.bases(r.qs + qb + x, 1, kind, ch);
      vcf_lit(s, "\t.\t.\t.");
      vcf_row_close(s, r, q_pos + x, 0, true);
      if (*kind) return;
    }
  } else if (len > r.svlen) { /* :719-813 */
    const bool ins = code == WGA_OP_I;
    if (tb == 0 || qb == 0 || (ins ? (tb > r.tn || qb + len > r.qn) : (tb + len > r.tn || qb > r.qn))) {
      if (*kind == 0u) *kind = 1u;
      return;
    }
    vcf_row_open(s, r, t_pos);
    s.bases(r.ts + tb - 1u, ins ? 1u : len + 1u, kind, ch);
    s.c((u8)'\t');
    s.bases(r.qs + qb - 1u, ins ? len + 1u : 1u, kind, ch);
    vcf_lit(s, "\t.\t.\t");
    if (r.neg) vcf_lit(s, "INV_NEST=TRUE;");
    if (ins)
      vcf_lit(s, "SVTYPE=INS;SVLEN=");
    else
      vcf_lit(s, "SVTYPE=DEL;SVLEN=");
    s.dec(len);
    vcf_lit(s, ";END=");
    s.dec(ins ? t_pos : t_pos + len);
    vcf_row_close(s, r, q_pos, ins ? q_pos + len : q_pos, false);
  }
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_paf_call_vcf(u32 n, const u32* __restrict__ ops, const u64* __restrict__ op_off,
                                                      const u8* __restrict__ strand_neg, u64 svlen,
                                                      const u64* __restrict__ ev, const u64* __restrict__ ev_off,
                                                      const wga_vcf_rec_dev* __restrict__ recs,
                                                      const u8* __restrict__ names, const u8* __restrict__ t_pool,
                                                      const u8* __restrict__ q_pool, u64* nbytes, wga_vcf_err_dev* err,
                                                      u8* out, const u64* __restrict__ out_off) {
  __shared__ u32x4_a16 s_text[4][FILL ? (WGA_VCF_TB + 32u) / 16u : 1u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 k = (u64)blockIdx.x * 4 + wave;
  if (k >= n) return;
  u8* const tbuf = (u8*)s_text[wave];
  const wga_vcf_rec_dev rr = recs[k];
  VcfRecCtx r;
  r.t_name = names + rr.t_name_off, r.q_name = names + rr.q_name_off;
  r.t_name_len = rr.t_name_len, r.q_name_len = rr.q_name_len;
  r.ts = t_pool + rr.t_off, r.qs = q_pool + rr.q_off;
  r.tn = rr.t_len, r.qn = rr.q_len;
  r.t_start = rr.t_start, r.t_end = rr.t_end, r.q_start = rr.q_start, r.q_end = rr.q_end;
  r.svlen = svlen;
  r.rops = ops + op_off[k], r.nops = op_off[k + 1] - op_off[k];
  r.ev = ev + 3u * ev_off[k];
  r.neg = strand_neg[k] != 0;
  const u64 n_items = 1u + (ev_off[k + 1] - ev_off[k]);
  u64 run = 0; /* bytes of the steps in front */
  for (u64 base = 0; base < n_items; base += 64u) {
    const u64 item = base + lane;
    VcfCount cs;
    cs.n = 0;
    u32 kind = 0, ch = 0;
    if (item < n_items) vcf_item(cs, r, item, &kind, &ch);
    const u64 bad = __ballot(kind != 0u);
    const u32 first_bad = bad ? (u32)__ffsll((unsigned long long)bad) - 1u : 64u;
    const u64 mine = lane < first_bad ? cs.n : 0ull; /* the record ends in front of its first bad item */
    const u64 inc = wave_incl_scan_u64(mine, lane);
    const u64 total = WGA_UNI64(__shfl((long long)inc, 63));
    if (FILL && total) { /* wave-uniform */
      u8* const g0 = out + out_off[k] + run;
      const u32 a = (u32)((uintptr_t)g0 & 15u);
      const bool staged = total <= (u64)WGA_VCF_TB;
      if (mine) {
        VcfEmit es;
        es.p = (staged ? tbuf + a : g0) + (inc - mine);
        vcf_item(es, r, item, &kind, &ch);
      }
      if (staged) {
        WGA_WAVE_SYNC();
        lds_text_flush(tbuf, a, (u32)total, g0 - a, lane, 64u);
        WGA_WAVE_SYNC();
      }
    }
    run += total;
    if (bad) {
      if (!FILL && lane == first_bad) {
        wga_vcf_err_dev e2;
        e2.item = item, e2.kind = kind, e2.ch = ch;
        err[k] = e2;
      }
      break;
    }
  }
  if (!FILL && lane == 0) nbytes[k] = run;
}

/* ============================================================================================ */
/* K17: BGZF blocks inflated where the text will be read ("bgzf reader on the  */
/*      GPU-direct path"; the reference reaches bgzf through htslib, converter.rs:183-184)        */
/* ============================================================================================ */
/* A BGZF file is a series of gzip members of at most 64 KiB, each an independent raw DEFLATE stream (RFC 1951) whose
 * uncompressed size stands in its trailer: the host walks the member headers (a hop per block), the compressed file goes
 * to the device as it is, and every block is inflated by one wave straight to its place in the text.  The decoder is
 * RFC 1951's own description — stored, fixed and dynamic blocks; canonical Huffman codes decoded bit by bit against the
 * per-length counts (3.2.2), length / distance symbols with their extra bits (3.2.5), the code-length alphabet (3.2.7) —
 * run by the wave's first lane with its tables in LDS; the parallelism is the tens of thousands of blocks of a genome.
 * Status per block: 0, or what was wrong (a corrupt stream never writes outside its block's output range). */
typedef uint16_t u16;
struct wga_bgzf_block_dev { /* = wga_bgzf_block */
  u64 in_off;  /* the raw DEFLATE stream inside the file image */
  u32 in_len;
  u32 out_len; /* ISIZE */
  u64 out_off;
};
#define WGA_INF_OK 0u
#define WGA_INF_INPUT 1u   /* the stream ends inside a symbol */
#define WGA_INF_TYPE 2u    /* block type 3, or a stored block whose length check fails */
#define WGA_INF_CODES 3u   /* over-subscribed / missing code lengths, a repeat without a previous length */
#define WGA_INF_SYMBOL 4u  /* a bit pattern no code has, length / distance symbol out of range */
#define WGA_INF_DIST 5u    /* a distance beyond the start of the block's output */
#define WGA_INF_SIZE 6u    /* more or fewer bytes than ISIZE */

static __device__ const u16 k_inf_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static __device__ const u8 k_inf_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static __device__ const u16 k_inf_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static __device__ const u8 k_inf_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static __device__ const u8 k_inf_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* The bit stream: `buf` holds the next cnt bits; the four bytes behind them were loaded when the four before were taken
 * (`ahead`), so that the load's latency passes under the decoding of the bits already there; pos = the next byte to load. */
typedef u32 u32_a1 __attribute__((aligned(1)));
struct InfBits {
  const u8* in;
  u32 pos, end;
  u64 buf;
  u32 cnt, err;
  u32 ahead, have_ahead;
};
__device__ __forceinline__ void inf_prime(InfBits& b) { /* start (or restart) the look-ahead at pos */
  b.have_ahead = 0u;
  if (b.pos + 4u <= b.end) {
    b.ahead = *(const u32_a1*)(b.in + b.pos);
    b.pos += 4u;
    b.have_ahead = 1u;
  }
}
__device__ __forceinline__ void inf_refill(InfBits& b) {
  if (b.cnt <= 32u && b.have_ahead) {
    b.buf |= (u64)b.ahead << b.cnt;
    b.cnt += 32u;
    inf_prime(b);
  }
}
__device__ __forceinline__ u32 inf_bits(InfBits& b, u32 n) { /* n <= 16 bits, least significant first (3.1.1) */
  while (b.cnt < n) {
    if (b.have_ahead) {
      inf_refill(b);
      continue;
    }
    if (b.pos >= b.end) { /* the last one to three bytes come one by one */
      b.err = b.err ? b.err : WGA_INF_INPUT;
      return 0u;
    }
    b.buf |= (u64)b.in[b.pos++] << b.cnt;
    b.cnt += 8u;
  }
  const u32 v = (u32)b.buf & ((1u << n) - 1u);
  b.buf >>= n;
  b.cnt -= n;
  return v;
}
/* a canonical code: count[len] codes of every length, the symbols in code order */
struct InfCode {
  u16* count;  /* [16] */
  u16* symbol; /* [n]  */
};
/* lengths -> the code; returns < 0 for an over-subscribed set, > 0 for an incomplete one, 0 for a complete one */
__device__ __forceinline__ int inf_construct(InfCode h, const u16* length, u32 n) {
  u16 offs[16];
  for (u32 l = 0; l < 16u; l++) h.count[l] = 0;
  for (u32 k = 0; k < n; k++) h.count[length[k]]++;
  if (h.count[0] == n) return 0; /* no codes: complete, and never decoded */
  int left = 1;
  for (u32 l = 1; l < 16u; l++) {
    left <<= 1;
    left -= (int)h.count[l];
    if (left < 0) return left;
  }
  offs[1] = 0;
  for (u32 l = 1; l < 15u; l++) offs[l + 1] = (u16)(offs[l] + h.count[l]);
  for (u32 k = 0; k < n; k++)
    if (length[k]) h.symbol[offs[length[k]]++] = (u16)k;
  return left;
}
/* first-level table of a code: the WGA_INF_FAST bits in front of the stream, as they lie there (a code's bits arrive most
 * significant first, 3.1.1: the index is the code reversed, and every pattern behind it), -> symbol << 4 | length; 0 for
 * patterns whose code is longer (decoded bit by bit) or that no code has */
#define WGA_INF_FAST 9u
__device__ __forceinline__ void inf_fast_table(InfCode h, u16* fast) {
  for (u32 k = 0; k < (1u << WGA_INF_FAST); k++) fast[k] = 0;
  u32 code = 0, index = 0;
  for (u32 l = 1; l <= WGA_INF_FAST; l++) {
    code <<= 1;
    const u32 cnt = h.count[l];
    for (u32 j = 0; j < cnt; j++, code++, index++) {
      u32 rev = 0;
      for (u32 t = 0; t < l; t++) rev |= ((code >> t) & 1u) << (l - 1u - t);
      const u16 e = (u16)((u32)h.symbol[index] << 4 | l);
      for (u32 x = rev; x < (1u << WGA_INF_FAST); x += 1u << l) fast[x] = e;
    }
  }
}
__device__ __forceinline__ int inf_decode(InfBits& b, InfCode h);
/* one symbol through the first-level table when the stream holds enough bits for it, else bit by bit */
__device__ __forceinline__ int inf_decode_fast(InfBits& b, InfCode h, const u16* fast) {
  inf_refill(b);
  if (b.cnt >= WGA_INF_FAST) {
    const u32 e = fast[(u32)b.buf & ((1u << WGA_INF_FAST) - 1u)];
    if (e) {
      const u32 l = e & 15u;
      b.buf >>= l;
      b.cnt -= l;
      return (int)(e >> 4);
    }
  }
  return inf_decode(b, h);
}
__device__ __forceinline__ int inf_decode(InfBits& b, InfCode h) { /* one symbol, or -1 */
  int code = 0, first = 0, index = 0;
  for (u32 l = 1; l < 16u; l++) {
    code |= (int)inf_bits(b, 1u);
    if (b.err) return -1;
    const int cnt = (int)h.count[l];
    if (code - cnt < first) return (int)h.symbol[index + (code - first)];
    index += cnt;
    first += cnt;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

__global__ __launch_bounds__(256) void k_bgzf_inflate(const u8* __restrict__ in, u64 in_bytes, u32 n_blocks,
                                                      const wga_bgzf_block_dev* __restrict__ blocks, u8* out, u32* status) {
  __shared__ u16 s_lencnt[4][16], s_lensym[4][288], s_distcnt[4][16], s_distsym[4][32], s_len[4][320], s_fast[4][1u << WGA_INF_FAST];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 k = (u64)blockIdx.x * 4 + wave;
  if (k >= n_blocks || lane != 0u) return; /* one lane decodes; the wave's other lanes have nothing to do */
  const wga_bgzf_block_dev blk = blocks[k];
  InfBits b;
  b.in = in + blk.in_off;
  b.pos = 0, b.end = blk.in_off + blk.in_len <= in_bytes ? blk.in_len : 0u;
  b.buf = 0, b.cnt = 0, b.err = 0;
  inf_prime(b);
  u8* const o = out + blk.out_off;
  const u32 cap = blk.out_len;
  u32 n_out = 0, err = 0;
  InfCode lencode, distcode;
  lencode.count = s_lencnt[wave], lencode.symbol = s_lensym[wave];
  distcode.count = s_distcnt[wave], distcode.symbol = s_distsym[wave];
  u16* const length = s_len[wave];
  u16* const fast = s_fast[wave];
  u32 last = 0;
  while (!last && !err && !b.err) {
    last = inf_bits(b, 1u);
    const u32 type = inf_bits(b, 2u);
    if (b.err) break;
    if (type == 0u) { /* stored (3.2.4): to the next byte boundary, LEN, NLEN, the bytes */
      b.pos -= (b.cnt >> 3) + (b.have_ahead ? 4u : 0u); /* whole bytes that were fetched ahead go back */
      b.buf = 0, b.cnt = 0, b.have_ahead = 0u;
      if (b.pos + 4u > b.end) {
        err = WGA_INF_INPUT;
        break;
      }
      const u32 len = (u32)b.in[b.pos] | ((u32)b.in[b.pos + 1] << 8), nlen = (u32)b.in[b.pos + 2] | ((u32)b.in[b.pos + 3] << 8);
      b.pos += 4u;
      if (len != (~nlen & 0xFFFFu)) {
        err = WGA_INF_TYPE;
        break;
      }
      if (b.pos + len > b.end) {
        err = WGA_INF_INPUT;
        break;
      }
      if (n_out + len > cap) {
        err = WGA_INF_SIZE;
        break;
      }
      for (u32 j = 0; j < len; j++) o[n_out + j] = b.in[b.pos + j];
      n_out += len;
      b.pos += len;
      inf_prime(b);
      continue;
    }
    if (type == 3u) {
      err = WGA_INF_TYPE;
      break;
    }
    if (type == 1u) { /* fixed codes (3.2.6) */
      u32 sym = 0;
      for (; sym < 144u; sym++) length[sym] = 8;
      for (; sym < 256u; sym++) length[sym] = 9;
      for (; sym < 280u; sym++) length[sym] = 7;
      for (; sym < 288u; sym++) length[sym] = 8;
      (void)inf_construct(lencode, length, 288u);
      for (sym = 0; sym < 30u; sym++) length[sym] = 5;
      (void)inf_construct(distcode, length, 30u);
    } else { /* dynamic codes (3.2.7) */
      const u32 nlen = inf_bits(b, 5u) + 257u, ndist = inf_bits(b, 5u) + 1u, ncode = inf_bits(b, 4u) + 4u;
      if (b.err) break;
      if (nlen > 286u || ndist > 30u) {
        err = WGA_INF_CODES;
        break;
      }
      u32 idx = 0;
      for (; idx < ncode; idx++) length[k_inf_clorder[idx]] = (u16)inf_bits(b, 3u);
      for (; idx < 19u; idx++) length[k_inf_clorder[idx]] = 0;
      if (b.err) break;
      if (inf_construct(lencode, length, 19u) != 0) { /* the code-length code must be complete */
        err = WGA_INF_CODES;
        break;
      }
      idx = 0;
      while (idx < nlen + ndist && !err) {
        const int sym = inf_decode(b, lencode);
        if (sym < 0) {
          err = b.err ? b.err : WGA_INF_SYMBOL;
          break;
        }
        if (sym < 16) {
          length[idx++] = (u16)sym;
        } else {
          u32 prev = 0, rep;
          if (sym == 16) {
            if (idx == 0u) {
              err = WGA_INF_CODES;
              break;
            }
            prev = length[idx - 1u];
            rep = 3u + inf_bits(b, 2u);
          } else if (sym == 17) {
            rep = 3u + inf_bits(b, 3u);
          } else {
            rep = 11u + inf_bits(b, 7u);
          }
          if (b.err) break;
          if (idx + rep > nlen + ndist) {
            err = WGA_INF_CODES;
            break;
          }
          while (rep--) length[idx++] = (u16)prev;
        }
      }
      if (err || b.err) break;
      if (length[256] == 0u) { /* no end-of-block code */
        err = WGA_INF_CODES;
        break;
      }
      /* an incomplete set is only allowed when it holds a single code (zlib's rule) */
      int left = inf_construct(lencode, length, nlen);
      if (left < 0 || (left > 0 && nlen - lencode.count[0] != 1u)) {
        err = WGA_INF_CODES;
        break;
      }
      left = inf_construct(distcode, length + nlen, ndist);
      if (left < 0 || (left > 0 && ndist - distcode.count[0] != 1u)) {
        err = WGA_INF_CODES;
        break;
      }
    }
    /* the block's symbols (3.2.3, 3.2.5) */
    inf_fast_table(lencode, fast);
    for (;;) {
      int sym = inf_decode_fast(b, lencode, fast);
      if (sym < 0) {
        err = b.err ? b.err : WGA_INF_SYMBOL;
        break;
      }
      if (sym < 256) {
        if (n_out >= cap) {
          err = WGA_INF_SIZE;
          break;
        }
        o[n_out++] = (u8)sym;
        continue;
      }
      if (sym == 256) break;
      sym -= 257;
      if (sym >= 29) {
        err = WGA_INF_SYMBOL;
        break;
      }
      const u32 len = (u32)k_inf_lbase[sym] + inf_bits(b, (u32)k_inf_lext[sym]);
      const int ds = inf_decode(b, distcode);
      if (ds < 0 || ds >= 30) {
        err = b.err ? b.err : WGA_INF_SYMBOL;
        break;
      }
      const u32 dist = (u32)k_inf_dbase[ds] + inf_bits(b, (u32)k_inf_dext[ds]);
      if (b.err) break;
      if (dist > n_out) {
        err = WGA_INF_DIST;
        break;
      }
      if (n_out + len > cap) {
        err = WGA_INF_SIZE;
        break;
      }
      for (u32 j = 0; j < len; j++) o[n_out + j] = o[n_out + j - dist]; /* byte by byte: the ranges may overlap */
      n_out += len;
    }
  }
  if (!err) err = b.err;
  if (!err && n_out != cap) err = WGA_INF_SIZE;
  status[k] = err;
}

#endif /* WGA_KERNELS3_H */
