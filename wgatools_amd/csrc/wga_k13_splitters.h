/*
 * wga_k13_splitters.h — K13 / K14: the PAF and MAF line splitters (paf.rs:24-30,50-78; maf.rs:25-36,138-211).
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K13_SPLITTERS_H
#define WGA_K13_SPLITTERS_H

#include "wga_kernels.h"

/* ============================================================================================ */
/* K13: PAF field splitter (the csv / serde floor of paf.rs:24-30,50-78)   */
/* ============================================================================================ */
/* The file text goes to the device once; the bytes that can end a field or a record — tab, newline,
 * and the two that need the csv crate's full state machine, '"' and '\r' — are listed in order
 * (count, scan, fill over 4 KB blocks), with the newlines' ranks in that list.  Then one thread per
 * line reads its dozen-odd delimiters: the 12 fixed fields (u64::from_str, Strand), the two name
 * spans and the span of the cg:Z: text, which the tokeniser reads in place.  Anything outside the
 * plain case — a quote or CR on the line, fewer than 12 fields, a bad integer or strand, a cs:Z:
 * tag standing in for cg:Z: — marks the line WGA_PAF_FALLBACK and the caller re-reads the file with
 * its csv-semantics parser (same records, or the reference's exact error).  4 B read twice per byte
 * of text; the per-line pass is negligible. */
#define WGA_PAF_OK 0
#define WGA_PAF_SKIP 1     /* blank line or '#' comment (csv reader: skipped) */
#define WGA_PAF_FALLBACK 2
struct wga_paf_line_dev {
  u64 num[9]; /* query_length, query_start, query_end, target_length, target_start, target_end, matches, block_length, mapq */
  u64 qname_off, tname_off, cg_beg, cg_end; /* byte offsets in the text; cg_beg == WGA_NONE: no cg:Z: tag */
  u32 qname_len, tname_len, n_fields;
  u8 strand_neg, status, pad[2];
};

/* MODE 0 (PAF): tab, newline, '"', CR.  MODE 1 (MAF): newline, the ASCII white space of
 * split_whitespace (9-13, 32) and every byte >= 0x80 (Unicode white space: left to the host). */
template <int MODE>
__device__ __forceinline__ u32 paf_delim_masks(const u32 w[4], u32* nl_mask) {
  u32 dm = 0, nm = 0;
#pragma unroll
  for (int d = 0; d < 4; d++) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const u32 ch = (w[d] >> (8 * b)) & 0xFFu;
      const u32 bit = 1u << (4 * d + b);
      nm |= ch == 0x0Au ? bit : 0u;
      if (MODE == 0)
        dm |= (ch == 0x0Au || ch == 0x09u || ch == 0x22u || ch == 0x0Du) ? bit : 0u;
      else
        dm |= (ch - 9u <= 4u || ch == 0x20u || ch >= 0x80u) ? bit : 0u;
    }
  }
  *nl_mask = nm;
  return dm;
}

template <bool FILL, int MODE>
__global__ __launch_bounds__(256) void k_paf_delims(const u8* __restrict__ text, u64 n_bytes, u64* blk,
                                                    const u64* blk_off, u64* delims, u64* nl_idx) {
  __shared__ u64 s_w[5];
  const u64 c = ((u64)blockIdx.x * 256u + threadIdx.x) * 16u;
  u32 w[4] = {0, 0, 0, 0};
  if (c + 16u <= n_bytes) {
    const u32x4_a1 a = *(const u32x4_a1*)(text + c);
    w[0] = a[0], w[1] = a[1], w[2] = a[2], w[3] = a[3];
  } else if (c < n_bytes) {
    for (u32 j = 0; j < (u32)(n_bytes - c); j++) w[j >> 2] |= (u32)text[c + j] << (8u * (j & 3u));
  }
  u32 nm;
  u32 dm = paf_delim_masks<MODE>(w, &nm);
  const u64 cnt = (u64)__builtin_popcount(dm) | ((u64)__builtin_popcount(nm) << 32);
  u64 tot;
  const u64 ex = block_excl_scan_u64(cnt, s_w, &tot);
  if (!FILL) {
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
    return;
  }
  const u64 base = blk_off[blockIdx.x] + ex; /* both halves advance independently: totals < 2^32 */
  u64 di = base & 0xFFFFFFFFull, ni = base >> 32;
  while (dm) {
    const u32 j = (u32)__builtin_ctz(dm);
    dm &= dm - 1u;
    delims[di] = c + j;
    if ((nm >> j) & 1u) nl_idx[ni++] = di;
    di++;
  }
}

/* u64::from_str on text[a, b): optional '+', >= 1 digit, no overflow */
__device__ __forceinline__ bool paf_parse_u64(const u8* __restrict__ text, u64 a, u64 b, u64* out) {
  if (a < b && text[a] == (u8)'+') a++;
  if (a >= b || b - a > 20u) return false;
  u64 v = 0;
  for (u64 k = a; k < b; k++) {
    const u32 d = (u32)text[k] - 0x30u;
    if (d > 9u) return false;
    if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10ull) return false;
    v = v * 10ull + d;
  }
  *out = v;
  return true;
}

__global__ __launch_bounds__(256) void k_paf_fields(const u8* __restrict__ text, u64 n_bytes, u64 n_lines,
                                                    u64 n_newlines, u64 n_delims,
                                                    const u64* __restrict__ delims,
                                                    const u64* __restrict__ nl_idx, wga_paf_line_dev* lines) {
  const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
  if (j >= n_lines) return;
  /* line j = bytes [s, e), its inner delimiters = delims[d0, d1) */
  const u64 d0 = j ? nl_idx[j - 1] + 1 : 0;
  const u64 s = j ? delims[nl_idx[j - 1]] + 1 : 0;
  const u64 d1 = j < n_newlines ? nl_idx[j] : n_delims;
  const u64 e = j < n_newlines ? delims[nl_idx[j]] : n_bytes;
  wga_paf_line_dev L;
  for (int k = 0; k < 9; k++) L.num[k] = 0;
  L.qname_off = L.tname_off = 0;
  L.cg_beg = L.cg_end = WGA_NONE;
  L.qname_len = L.tname_len = L.n_fields = 0;
  L.strand_neg = 0;
  L.pad[0] = L.pad[1] = 0;
  u8 status = WGA_PAF_OK;
  if (s == e || text[s] == (u8)'#') {
    status = WGA_PAF_SKIP;
    for (u64 d = d0; d < d1 && status == WGA_PAF_SKIP; d++) /* a '\r' inside would end the comment for csv */
      if (text[delims[d]] == 0x0Du) status = WGA_PAF_FALLBACK;
  } else {
    u64 fs = s; /* start of the current field */
    u32 nf = 0;
    bool seen_cg = false, seen_cs = false;
    for (u64 d = d0; d <= d1 && status == WGA_PAF_OK; d++) {
      u64 fe = e;
      if (d < d1) {
        fe = delims[d];
        if (text[fe] != 0x09u) { /* a quote or CR: the csv state machine decides */
          status = WGA_PAF_FALLBACK;
          break;
        }
      }
      bool ok = true;
      switch (nf) {
        case 0: L.qname_off = fs; L.qname_len = (u32)(fe - fs); ok = fe - fs < 0xFFFFFFFFull; break;
        case 1: ok = paf_parse_u64(text, fs, fe, &L.num[0]); break;
        case 2: ok = paf_parse_u64(text, fs, fe, &L.num[1]); break;
        case 3: ok = paf_parse_u64(text, fs, fe, &L.num[2]); break;
        case 4:
          ok = fe - fs == 1u && (text[fs] == (u8)'+' || text[fs] == (u8)'-');
          L.strand_neg = ok && text[fs] == (u8)'-' ? 1 : 0;
          break;
        case 5: L.tname_off = fs; L.tname_len = (u32)(fe - fs); ok = fe - fs < 0xFFFFFFFFull; break;
        case 6: ok = paf_parse_u64(text, fs, fe, &L.num[3]); break;
        case 7: ok = paf_parse_u64(text, fs, fe, &L.num[4]); break;
        case 8: ok = paf_parse_u64(text, fs, fe, &L.num[5]); break;
        case 9: ok = paf_parse_u64(text, fs, fe, &L.num[6]); break;
        case 10: ok = paf_parse_u64(text, fs, fe, &L.num[7]); break;
        case 11: ok = paf_parse_u64(text, fs, fe, &L.num[8]); break;
        default: /* tags: the first cg:Z: wins (paf.rs:126-130), a cs:Z: is only used without one */
          if (fe - fs >= 5u && text[fs] == (u8)'c' && text[fs + 2] == (u8)':' && text[fs + 3] == (u8)'Z' &&
              text[fs + 4] == (u8)':') {
            if (text[fs + 1] == (u8)'g' && !seen_cg) {
              seen_cg = true;
              L.cg_beg = fs + 5u;
              L.cg_end = fe;
            } else if (text[fs + 1] == (u8)'s') {
              seen_cs = true;
            }
          }
          break;
      }
      if (!ok) status = WGA_PAF_FALLBACK;
      nf++;
      fs = fe + 1u;
    }
    L.n_fields = nf;
    if (status == WGA_PAF_OK && (nf < 12u || (!seen_cg && seen_cs))) status = WGA_PAF_FALLBACK;
  }
  L.status = status;
  lines[j] = L;
}

/* ============================================================================================ */
/* K14: MAF line splitter (the reader of maf.rs:25-36,138-211,371-421 for plain files)           */
/* ============================================================================================ */
/* Same two lists as K13, with white space as the field delimiter.  One thread per line: a line that
 * starts with 's' (and is not the file's first line, which is always the header) is an s-line:
 * seven white-space separated tokens — mode, name, start, size, strand, srcSize, text — of which
 * the name and the text stay where they are (spans).  The K3 / K4 walks then read the rows straight
 * out of the uploaded file.  Blocks (maximal runs of s-lines) are put together by the caller. */
#define WGA_MAF_SLINE 0
#define WGA_MAF_OTHER 1    /* header, or a line that does not start with 's': ends a block */
#define WGA_MAF_FALLBACK 2 /* not seven tokens, a bad number or strand, a non-ASCII byte in front of the text */
struct wga_maf_line_dev {
  u64 num[3]; /* start, align_size, size (maf.rs:65-73) */
  u64 name_off, seq_off, seq_len;
  u32 name_len;
  u8 strand_neg, status, pad[2];
};

__global__ __launch_bounds__(256) void k_maf_lines(const u8* __restrict__ text, u64 n_bytes, u64 n_lines,
                                                   u64 n_newlines, u64 n_delims,
                                                   const u64* __restrict__ delims,
                                                   const u64* __restrict__ nl_idx, wga_maf_line_dev* lines) {
  const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
  if (j >= n_lines) return;
  const u64 d0 = j ? nl_idx[j - 1] + 1 : 0;
  const u64 s = j ? delims[nl_idx[j - 1]] + 1 : 0;
  const u64 d1 = j < n_newlines ? nl_idx[j] : n_delims;
  const u64 e = j < n_newlines ? delims[nl_idx[j]] : n_bytes;
  wga_maf_line_dev L;
  L.num[0] = L.num[1] = L.num[2] = 0;
  L.name_off = L.seq_off = L.seq_len = 0;
  L.name_len = 0;
  L.strand_neg = 0;
  L.pad[0] = L.pad[1] = 0;
  u8 status = WGA_MAF_OTHER;
  if (j > 0 && s < e && text[s] == (u8)'s') {
    status = WGA_MAF_SLINE;
    u64 prev = s;
    u32 nt = 0;
    for (u64 d = d0; d <= d1 && status == WGA_MAF_SLINE; d++) {
      const u64 p = d < d1 ? delims[d] : e;
      if (d < d1 && text[p] >= 0x80u) status = WGA_MAF_FALLBACK;
      if (p > prev) { /* a token */
        bool ok = true;
        switch (nt) {
          case 0: break; /* mode: its first char, not looked at again */
          case 1: L.name_off = prev; L.name_len = (u32)(p - prev); ok = p - prev < 0xFFFFFFFFull; break;
          case 2: ok = paf_parse_u64(text, prev, p, &L.num[0]); break;
          case 3: ok = paf_parse_u64(text, prev, p, &L.num[1]); break;
          case 4:
            ok = p - prev == 1u && (text[prev] == (u8)'+' || text[prev] == (u8)'-');
            L.strand_neg = ok && text[prev] == (u8)'-' ? 1 : 0;
            break;
          case 5: ok = paf_parse_u64(text, prev, p, &L.num[2]); break;
          case 6: L.seq_off = prev; L.seq_len = p - prev; break;
          default: ok = false; break; /* SurplusField */
        }
        if (!ok) status = WGA_MAF_FALLBACK;
        nt++;
      }
      prev = p + 1;
    }
    if (status == WGA_MAF_SLINE && nt != 7u) status = WGA_MAF_FALLBACK;
  }
  L.status = status;
  lines[j] = L;
}

#endif /* WGA_K13_SPLITTERS_H */
