/*
 * wga_k8_tokenise.h — K8: CIGAR text -> packed ops on the device (cigar.rs:43-75).
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K8_TOKENISE_H
#define WGA_K8_TOKENISE_H

#include "wga_kernels.h"

/* ============================================================================================ */
/* K8: CIGAR text -> packed ops on the device                                                   */
/* ============================================================================================ */
/* Same result as the host packer wga_cigar_pack (the nom tokeniser of cigar.rs:43-75 +
 * utils.rs:69-74), one wave per record.  Fast pass: the text goes through LDS 1 KiB at a time (16
 * bytes per lane); every non-digit byte is an op whose length is the digit run right before it,
 * parsed backwards from the staged text; op slots come from a wave scan of the per-lane op
 * counts.  Anything the fast pass does not cover — a token that is not "1..19 digits + one ASCII
 * char" (empty length, multi-char or multi-byte op, u64 overflow candidates), a length >= 2^28
 * (split into pieces), text that is empty or ends in digits — flags the record, and lane 0 then
 * re-tokenises it serially with the packer's exact rules and error reporting.  Records with
 * errors abort the run on the host anyway; the serial path only has to be right. */
struct wga_tok_err_dev {
  int err;      /* wga_rec_err */
  u32 tok_len;  /* offending token: length ... */
  u64 tok_off;  /* ... and offset inside the record's text */
};

__device__ __forceinline__ u32 tok_op_code(u8 c) {
  switch (c) {
    case 'M': return WGA_OP_M;
    case 'I': return WGA_OP_I;
    case 'D': return WGA_OP_D;
    case 'N': return WGA_OP_N;
    case 'S': return WGA_OP_S;
    case 'H': return WGA_OP_H;
    case 'P': return WGA_OP_P;
    case '=': return WGA_OP_EQ;
    case 'X': return WGA_OP_X;
    default: return WGA_OP_OTHER;
  }
}
__device__ __forceinline__ u32 tok_utf8_len(u8 c) {
  if (c < 0x80u) return 1u;
  if ((c >> 5) == 0x6u) return 2u;
  if ((c >> 4) == 0xEu) return 3u;
  if ((c >> 3) == 0x1Eu) return 4u;
  return 1u;
}

/* the packer's loop, verbatim in behaviour: returns the op count, writes ops if out != NULL */
__device__ __forceinline__ u64 tok_serial(const u8* text, u64 len, u32* out, wga_tok_err_dev* err) {
  u64 n = 0, p = 0;
  int e = WGA_REC_OK;
  u64 eoff = 0, elen = 0;
  if (len == 0) e = WGA_REC_PANIC;
  while (p < len) {
    const u64 ls = p;
    while (p < len && (u8)(text[p] - (u8)'0') < 10u) p++;
    const u64 ln = p - ls, os = p;
    while (p < len && (u8)(text[p] - (u8)'0') >= 10u) p++;
    const u64 on = p - os;
    if (on == 0 || tok_utf8_len(text[os]) < on) {
      e = WGA_REC_CIGAR_OP_INVALID;
      eoff = os;
      elen = on;
      break;
    }
    if (ln == 0) {
      e = WGA_REC_PARSE_INT;
      eoff = ls;
      elen = 0;
      break;
    }
    u64 v = 0;
    bool ovf = false;
    for (u64 k = 0; k < ln; k++) {
      const u64 d = (u64)(text[ls + k] - (u8)'0');
      if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10ull) {
        ovf = true;
        break;
      }
      v = v * 10ull + d;
    }
    if (ovf) {
      e = WGA_REC_PARSE_INT;
      eoff = ls;
      elen = ln;
      break;
    }
    const u32 code = on == 1 ? tok_op_code(text[os]) : (u32)WGA_OP_OTHER;
    const u32 cont = code == WGA_OP_I ? (u32)WGA_OP_I_CONT : code == WGA_OP_D ? (u32)WGA_OP_D_CONT : code;
    bool first = true;
    do {
      const u64 piece = v > (u64)WGA_OP_MAX_LEN ? (u64)WGA_OP_MAX_LEN : v;
      if (out) out[n] = ((u32)piece << 4) | (first ? code : cont);
      n++;
      v -= piece;
      first = false;
    } while (v > 0);
  }
  if (err) {
    err->err = e;
    err->tok_len = (u32)elen;
    err->tok_off = eoff;
  }
  return n;
}

#define WGA_TOK_HIST 32u /* bytes of the previous chunk kept in front of the staged one */

__global__ __launch_bounds__(256) void k_cigar_tokenise(u32 n, const u8* __restrict__ text,
                                                        const u64* __restrict__ text_beg,
                                                        const u64* __restrict__ text_end,
                                                        u64* op_cnt, wga_tok_err_dev* errs,
                                                        u32* ops, const u64* op_off) {
  __shared__ __attribute__((aligned(16))) u8 s_txt[4][WGA_TOK_HIST + 1024u + 16u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 i = (u64)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const u8* rec = text + text_beg[i]; /* CSR texts: text_end = text_beg + 1; spans of a file: two arrays */
  const u64 len = text_end[i] - text_beg[i];
  u32* out = ops ? ops + op_off[i] : (u32*)0;
  /* a record that turns out to hold an error packs to fewer ops than its non-digit bytes: the
   * fast pass must stay inside the record's own slots (the serial pass then rewrites them) */
  const u64 limit = ops ? op_off[i + 1] - op_off[i] : 0ull;
  u8* const buf = s_txt[wave];
  u8* const cur = buf + WGA_TOK_HIST; /* cur[-k] = byte k before the chunk */
  bool viol = len == 0 || (u8)(rec[len ? len - 1 : 0] - (u8)'0') < 10u; /* empty, or ends in digits */
  u64 base = 0;
  if (lane < WGA_TOK_HIST / 4u) ((u32*)buf)[lane] = 0x30303030u; /* history before the text: all '0' is never read
                                                                    as part of a run because p == 0 is checked */
  for (u64 c0 = 0; c0 < len && !viol; c0 += 1024) {
    WGA_WAVE_SYNC();
    const u64 c = c0 + (u64)lane * 16u;
    const u32 nv = c >= len ? 0u : (len - c >= 16u ? 16u : (u32)(len - c));
    u32 w[4] = {0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u};
    if (nv == 16u) {
      const u32x4_a1 a = *(const u32x4_a1*)(rec + c);
      w[0] = a[0], w[1] = a[1], w[2] = a[2], w[3] = a[3];
    } else if (nv) {
      for (u32 j = 0; j < nv; j++) {
        w[j >> 2] &= ~(0xFFu << (8u * (j & 3u)));
        w[j >> 2] |= (u32)rec[c + j] << (8u * (j & 3u));
      }
    }
    {
      const u32x4_a16 wv = {w[0], w[1], w[2], w[3]};
      *(u32x4_a16*)(cur + lane * 16u) = wv;
    }
    WGA_WAVE_SYNC();
    /* non-digit bytes of this lane (padding beyond the text counts as digits) */
    u32 nd = 0; /* 16-bit mask */
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const u32 ch = (w[d] >> (8 * b)) & 0xFFu;
        nd |= ((ch - 0x30u) >= 10u ? 1u : 0u) << (4 * d + b);
      }
    }
    const u32 cnt = (u32)__builtin_popcount(nd);
    const u32 incl = wave_incl_scan_u32(cnt);
    u32 slot = incl - cnt;
    u32 m = nd;
    bool bad = false;
    while (m) {
      const u32 j = (u32)__builtin_ctz(m);
      m &= m - 1u;
      const int p = (int)(lane * 16u + j); /* position inside the chunk */
      const u8 ch = cur[p];
      /* digits right before p, backwards (at most 19 accepted) */
      u64 v = 0, mul = 1;
      u32 nd_run = 0;
      int k = p - 1;
      const int kmin = -(int)(c0 < (u64)WGA_TOK_HIST ? c0 : (u64)WGA_TOK_HIST); /* readable history */
      while (k >= kmin && (u8)(cur[k] - (u8)'0') < 10u && nd_run < 20u) {
        v += (u64)(cur[k] - (u8)'0') * mul;
        mul *= 10ull;
        nd_run++;
        k--;
      }
      /* not "1..19 digits + one ASCII char", or the run may go on beyond what is staged, or the
       * length needs splitting: leave the record to the serial path */
      if (nd_run == 0u || nd_run > 19u || ch >= 0x80u || (k < kmin && c0 + (u64)p > (u64)nd_run) ||
          v > (u64)WGA_OP_MAX_LEN)
        bad = true;
      if (out && !bad && base + slot < limit) out[base + slot] = ((u32)v << 4) | tok_op_code(ch);
      slot++;
    }
    viol = __ballot(bad) != 0ull;
    base += (u64)wave_last_u32(incl);
    /* keep the chunk's last WGA_TOK_HIST bytes in front of the next one */
    WGA_WAVE_SYNC();
    u32 hv = 0;
    if (lane < WGA_TOK_HIST / 4u) hv = ((const u32*)(cur + 1024u - WGA_TOK_HIST))[lane];
    WGA_WAVE_SYNC();
    if (lane < WGA_TOK_HIST / 4u) ((u32*)buf)[lane] = hv;
  }
  wga_tok_err_dev e;
  e.err = WGA_REC_OK;
  e.tok_len = 0;
  e.tok_off = 0;
  if (viol) { /* wave-uniform */
    if (lane == 0) base = tok_serial(rec, len, out, &e);
  }
  if (lane == 0) {
    if (op_cnt) op_cnt[i] = base;
    if (errs) errs[i] = e;
  }
}

#endif /* WGA_K8_TOKENISE_H */
