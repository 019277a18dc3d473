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

#include "wga_kernels2.h"

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

#endif /* WGA_KERNELS3_H */
