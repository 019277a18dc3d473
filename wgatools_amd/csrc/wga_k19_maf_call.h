/*
 * wga_k19_maf_call.h — K19: the rules and the VCF rows of `call` on MAF ("K4 ... stream-compact
 * events"), on the run list K4 leaves on the device.
 *
 * The reference cuts a block into SV-safe chunks (find_safe_chunk_boundary, caller.rs:159-219), makes a record of each
 * (create_chunk_record :221-265) and folds its columns, grouped by cigar_cat_ext_caller, into VCF records
 * (call_within_var :388-608); round 5 did all of that on the host from the downloaded run list.  Everything in it is a
 * function of RUNS, not of columns:
 *   * a gap segment of the boundary scan = a maximal series of I / D / W runs; its start is the end of the last '=' / X run in
 *     front of it (a running maximum over the runs), and the chunk ends behind the LAST segment of the proposed range that is
 *     at least `svlen` columns (from the chunk's start on) — the segment that crosses the proposed end counts with its part
 *     inside and carries the end to its own (:186-216);
 *   * `after_m` in front of a run = the nearest earlier run of the chunk that is not W is '=' or X (W runs change nothing,
 *     :447-463; every other run sets the flag) — two ballots and a count of leading zeros;
 *   * the target / query offsets in front of a run are K4's non-gap prefixes minus the chunk's; the base "in front" of an
 *     INS / DEL is the last column of that '=' / X run, read from the rows where they lie (the reference slices the gap-stripped
 *     sequences at the same characters, :480-501,537-555);
 *   * rows: one per column of an X run with `-s` (:570-603), one per I / D run longer than `svlen` behind an '=' / X run
 *     (:464-569), one <INV> row in front of a '-' block's chunk with `inv` (:423-440).
 * One wave per block; a chunk is three passes over its runs, 64 runs a step (boundary; the run the chunk ends in; the rows).
 * Text as K16 writes it (wga_kernels3.h): a step counts its rows' bytes, a wave scan places them, the fill pass assembles the
 * step in LDS and stores it in 16-byte groups.  Two-call protocol: bytes per block and the first bad base (noodles-vcf's parse
 * error: a REF / ALT character outside ACGTN in either case), then the text.  A block's text ends in front of the CHUNK that
 * holds its first bad base: the reference collects a chunk's records before it writes any of them (:137-141).
 */
#ifndef WGA_K19_MAF_CALL_H
#define WGA_K19_MAF_CALL_H

#include "wga_kernels3.h"

struct wga_maf_vcf_rec_dev { /* = wga_maf_vcf_rec (wga_hip.h) */
  u64 t_name_off, q_name_off; /* into `names` */
  u32 t_name_len, q_name_len;
  u64 t_start, q_start, q_size; /* the s lines' start fields and the query's source size (maf.rs:65-73) */
  u32 q_neg, pad;
};

__device__ __forceinline__ u64 wave_incl_scan_max_u64(u64 v, u32 lane) {
#pragma unroll
  for (u32 d = 1; d < 64; d <<= 1) {
    const u64 o = __shfl_up(v, d);
    if (lane >= d && o > v) v = o;
  }
  return v;
}
__device__ __forceinline__ bool maf_cls_gap(u32 c) { return c == 1u || c == 2u || c == 4u; }
__device__ __forceinline__ bool maf_cls_adv_t(u32 c) { return c == 0u || c == 3u || c == 2u; }
__device__ __forceinline__ bool maf_cls_adv_q(u32 c) { return c == 0u || c == 3u || c == 1u; }

struct MafVcfCtx {
  const u8 *t_name, *q_name, *trow, *qrow;
  u32 t_name_len, q_name_len;
  bool neg;
};
template <class S>
__device__ __forceinline__ void mvcf_open(S& s, const MafVcfCtx& r, u64 pos) { /* "<chro>\t<pos>\t.\t" */
  s.str(r.t_name, r.t_name_len);
  s.c((u8)'\t');
  s.dec(pos);
  vcf_lit(s, "\t.\t");
}
template <class S>
__device__ __forceinline__ void mvcf_close(S& s, const MafVcfCtx& r, u64 a, u64 b, bool three) { /* "\tGT:QI\t1|1:<query>@<a>[@<b>]@<P|N>\n" */
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
/* what a lane's run asks for: kind 0 nothing, 1 the <INV> row (lane 0's extra item), 2 SNP rows, 3 INS, 4 DEL */
struct MafVcfItem {
  u32 kind;
  u64 s0, len, pc; /* first column, columns, the column of the base in front (INS / DEL) or of the <INV> row's base */
  u64 t_pos, q_pos, q_end, t_end; /* offsets in front of the run (chunk coordinates added); <INV>: the chunk's */
};
template <class S>
__device__ __forceinline__ void mvcf_item(S& s, const MafVcfCtx& r, const MafVcfItem& it, u32* kind, u32* ch) {
  if (it.kind == 1u) { /* caller.rs:423-440 */
    mvcf_open(s, r, it.t_pos + 1u);
    s.bases(r.trow + it.pc, 1, kind, ch);
    vcf_lit(s, "\t<INV>\t.\t.\tSVTYPE=INV;END=");
    s.dec(it.t_end);
    mvcf_close(s, r, it.q_pos, it.q_end, false);
  } else if (it.kind == 2u) { /* :570-603, one row per column */
    for (u64 x = 0; x < it.len; x++) {
      mvcf_open(s, r, it.t_pos + x + 1u);
      s.bases(r.trow + it.s0 + x, 1, kind, ch);
      s.c((u8)'\t');
      s.bases(r.qrow + it.s0 + x, 1, kind, ch);
      vcf_lit(s, "\t.\t.\t.");
      mvcf_close(s, r, it.q_pos + x, 0, true);
      if (*kind) return;
    }
  } else if (it.kind == 3u || it.kind == 4u) { /* :464-569 */
    const bool ins = it.kind == 3u;
    mvcf_open(s, r, it.t_pos);
    s.bases(r.trow + it.pc, 1, kind, ch);
    if (!ins) s.bases(r.trow + it.s0, it.len, kind, ch);
    s.c((u8)'\t');
    s.bases(r.qrow + it.pc, 1, kind, ch);
    if (ins) s.bases(r.qrow + it.s0, it.len, kind, ch);
    vcf_lit(s, "\t.\t.\t");
    if (r.neg) vcf_lit(s, "INV_NEST=TRUE;");
    if (ins)
      vcf_lit(s, "SVTYPE=INS;SVLEN=");
    else
      vcf_lit(s, "SVTYPE=DEL;SVLEN=");
    s.dec(it.len);
    vcf_lit(s, ";END=");
    s.dec(ins ? it.t_pos : it.t_pos + it.len);
    mvcf_close(s, r, it.q_pos, ins ? it.q_pos + it.len : it.q_pos, false);
  }
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_maf_call_vcf(u32 n, const u8* __restrict__ rows, const u64* __restrict__ t_off,
                                                      const u64* __restrict__ q_off, const u64* __restrict__ cols,
                                                      const u64* __restrict__ runs, const u64* __restrict__ run_off,
                                                      const wga_maf_vcf_rec_dev* __restrict__ recs,
                                                      const u8* __restrict__ names, u32 snp, u32 inv, u64 svlen, u64 chunk,
                                                      u64* nbytes, wga_vcf_err_dev* err, u8* out,
                                                      const u64* __restrict__ out_off) {
  __shared__ u32x4_a16 s_text[4][FILL ? (WGA_VCF_TB + 32u) / 16u : 1u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 k = (u64)blockIdx.x * 4 + wave;
  if (k >= n) return;
  u8* const tbuf = (u8*)s_text[wave];
  const wga_maf_vcf_rec_dev rr = recs[k];
  MafVcfCtx r;
  r.t_name = names + rr.t_name_off, r.q_name = names + rr.q_name_off;
  r.t_name_len = rr.t_name_len, r.q_name_len = rr.q_name_len;
  r.trow = rows + t_off[k], r.qrow = rows + q_off[k];
  r.neg = rr.q_neg != 0u;
  const u64* const R = runs + 3u * run_off[k];
  const u64 nr = run_off[k + 1] - run_off[k];
  const u64 total = cols[k];
  const u64 below = (1ull << lane) - 1ull;
  u64 written = 0;   /* bytes of the chunks in front */
  u64 cs = 0, rk = 0; /* the chunk's first column and the run it lies in */
  u32 bad_kind = 0, bad_ch = 0;
  u64 bad_item = 0, items_before = 0;
  while (cs < total && nr) { /* wave-uniform */
    const u64 proposed = chunk >= total - cs ? total : cs + chunk;
    /* ---- the chunk's end (find_safe_chunk_boundary on runs) ---- */
    u64 safe_end = proposed;
    {
      u64 last_m_end = cs; /* end of the last '=' / X run in front of the step = where a gap segment that is open began */
      for (u64 base = rk; base < nr; base += 64u) {
        const u64 j = base + lane;
        const bool valid = j < nr;
        const u64 w0 = valid ? R[3u * j] : 0ull, w1 = (valid && j + 1u < nr) ? R[3u * (j + 1u)] : 0ull;
        const u64 s = w0 >> 3, e = (valid && j + 1u < nr) ? (w1 >> 3) : total;
        const u32 c = (u32)(w0 & 7u);
        const bool gap = valid && maf_cls_gap(c);
        const bool next_gap = valid && j + 1u < nr && maf_cls_gap((u32)(w1 & 7u));
        const u64 mend = (valid && !gap) ? e : 0ull;
        const u64 incl = wave_incl_scan_max_u64(mend, lane);
        u64 seg = __shfl_up(incl, 1u);
        if (lane == 0u) seg = 0ull;
        if (seg < last_m_end) seg = last_m_end; /* the segment's first column */
        const u64 gs = seg > cs ? seg : cs;
        const bool seg_end = gap && !next_gap && seg < proposed; /* a segment the reference's scan reaches (its first run starts in front of the proposed end) */
        const bool q1 = seg_end && e < proposed && e - gs >= svlen;
        const bool q2 = seg_end && e >= proposed && proposed - gs >= svlen;
        const u64 cand = q1 ? e : (e < total ? e : proposed);
        const u64 any = __ballot(q1 || q2);
        if (any) safe_end = __shfl(cand, 63 - (int)__builtin_clzll(any)); /* the last one in order decides */
        const u64 top = __shfl(incl, 63);
        if (top > last_m_end) last_m_end = top;
        /* on while the step's last run lies in front of the proposed end or inside a segment that began there */
        const bool relevant = valid && (s < proposed || (gap && seg < proposed));
        if (!((__ballot(relevant) >> 63) & 1ull)) break;
      }
    }
    const u64 ce = safe_end;
    /* ---- chunk coordinates (create_chunk_record): non-gap characters in front of cs and of ce ---- */
    u64 tb0, qb0, tb1, qb1, rk_next = nr;
    {
      const u64 w = R[3u * rk];
      const u32 c = (u32)(w & 7u);
      const u64 d = cs - (w >> 3);
      tb0 = R[3u * rk + 1u] + (maf_cls_adv_t(c) ? d : 0ull);
      qb0 = R[3u * rk + 2u] + (maf_cls_adv_q(c) ? d : 0ull);
    }
    if (ce < total) { /* the run ce lies in: the first one that ends behind it */
      for (u64 base = rk; base < nr; base += 64u) {
        const u64 j = base + lane;
        const bool valid = j < nr;
        const u64 e = valid ? (j + 1u < nr ? (R[3u * (j + 1u)] >> 3) : total) : 0ull;
        const u64 hit = __ballot(valid && e > ce);
        if (hit) {
          rk_next = base + (u64)(__ffsll((unsigned long long)hit) - 1);
          break;
        }
      }
    }
    {
      const u64 jl = rk_next < nr ? rk_next : nr - 1u;
      const u64 w = R[3u * jl];
      const u32 c = (u32)(w & 7u);
      const u64 d = (rk_next < nr ? ce : total) - (w >> 3);
      tb1 = R[3u * jl + 1u] + (maf_cls_adv_t(c) ? d : 0ull);
      qb1 = R[3u * jl + 2u] + (maf_cls_adv_q(c) ? d : 0ull);
    }
    const u64 t_align = tb1 - tb0, q_align = qb1 - qb0;
    const u64 t_start = rr.t_start + tb0, t_end = t_start + t_align, qss = rr.q_start + qb0;
    const u64 q_start = r.neg ? rr.q_size - qss - q_align : qss, q_end = r.neg ? rr.q_size - qss : qss + q_align;
    /* ---- the <INV> row's base: the first column of the chunk that holds a target base ---- */
    const bool want_inv = r.neg && t_align != 0ull && inv != 0u; /* wave-uniform */
    u64 inv_col = 0;
    if (want_inv) {
      for (u64 base = rk; base < nr; base += 64u) {
        const u64 j = base + lane;
        const bool valid = j < nr;
        const u64 w0 = valid ? R[3u * j] : 0ull;
        const u64 hit = __ballot(valid && maf_cls_adv_t((u32)(w0 & 7u)));
        if (hit) {
          const u64 s = __shfl(w0 >> 3, __ffsll((unsigned long long)hit) - 1);
          inv_col = s > cs ? s : cs;
          break;
        }
      }
    }
    /* ---- the rows ---- */
    u64 chunk_bytes = 0;
    bool carry_after = false; /* after_m in front of the step */
    u64 carry_pc = 0;         /* last column of the last '=' / X run in front of the step */
    bool first_step = true, chunk_bad = false;
    for (u64 base = rk; base < nr; base += 64u) {
      const u64 j = base + lane;
      const bool inr = j < nr;
      const u64 w0 = inr ? R[3u * j] : 0ull;
      const u64 s = w0 >> 3;
      const bool valid = inr && s < ce;
      if (__ballot(valid) == 0ull) break;
      const u64 e = inr ? (j + 1u < nr ? (R[3u * (j + 1u)] >> 3) : total) : 0ull;
      const u32 c = (u32)(w0 & 7u);
      const u64 s0 = s > cs ? s : cs, e0 = e < ce ? e : ce, len = valid ? e0 - s0 : 0ull;
      const u64 NW = __ballot(valid && c != 4u), MM = __ballot(valid && (c == 0u || c == 3u));
      const u64 nwb = NW & below;
      const int jl = nwb ? 63 - (int)__builtin_clzll(nwb) : 0;
      const bool after = nwb ? ((MM >> jl) & 1ull) != 0ull : carry_after;
      /* the base in front: last column of the nearest '=' / X run below (after_m says it is the nearest non-W run) */
      const u64 mb = MM & below;
      const u64 pe = __shfl(e0, mb ? 63 - (int)__builtin_clzll(mb) : 0);
      const u64 pc = mb ? pe - 1u : carry_pc;
      MafVcfItem it;
      it.kind = 0u, it.s0 = s0, it.len = len, it.pc = pc;
      it.t_pos = t_start + (inr ? R[3u * j + 1u] : 0ull) + (maf_cls_adv_t(c) ? s0 - s : 0ull) - tb0;
      it.q_pos = q_start + (inr ? R[3u * j + 2u] : 0ull) + (maf_cls_adv_q(c) ? s0 - s : 0ull) - qb0;
      it.q_end = 0, it.t_end = 0;
      if (valid) {
        if (c == 3u && snp)
          it.kind = 2u;
        else if ((c == 1u || c == 2u) && len > svlen && after)
          it.kind = c == 1u ? 3u : 4u;
      }
      MafVcfItem iv; /* lane 0 of the chunk's first step also holds the <INV> row, in front of its own */
      iv.kind = (first_step && want_inv && lane == 0u) ? 1u : 0u;
      iv.s0 = 0, iv.len = 0, iv.pc = inv_col, iv.t_pos = t_start, iv.t_end = t_end, iv.q_pos = q_start, iv.q_end = q_end;
      VcfCount cnt;
      cnt.n = 0;
      u32 kind = 0, ch = 0;
      if (iv.kind) mvcf_item(cnt, r, iv, &kind, &ch);
      if (it.kind && !kind) mvcf_item(cnt, r, it, &kind, &ch);
      const u64 bad = __ballot(kind != 0u);
      const u64 inc = wave_incl_scan_u64(cnt.n, lane);
      const u64 step_bytes = __shfl(inc, 63);
      if (bad) { /* the chunk's records are collected before any is written: the block's text ends in front of this chunk */
        const u32 fb = (u32)__ffsll((unsigned long long)bad) - 1u;
        bad_kind = __shfl(kind, (int)fb), bad_ch = __shfl(ch, (int)fb);
        bad_item = items_before + fb;
        chunk_bad = true;
        break;
      }
      if (FILL && step_bytes) { /* wave-uniform */
        u8* const g0 = out + out_off[k] + written + chunk_bytes;
        const u32 a = (u32)((uintptr_t)g0 & 15u);
        const bool staged = step_bytes <= (u64)WGA_VCF_TB;
        if (cnt.n) {
          VcfEmit es;
          es.p = (staged ? tbuf + a : g0) + (inc - cnt.n);
          if (iv.kind) mvcf_item(es, r, iv, &kind, &ch);
          if (it.kind) mvcf_item(es, r, it, &kind, &ch);
        }
        if (staged) {
          WGA_WAVE_SYNC();
          lds_text_flush(tbuf, a, (u32)step_bytes, g0 - a, lane, 64u);
          WGA_WAVE_SYNC();
        }
      }
      chunk_bytes += step_bytes;
      items_before += 64u;
      if (NW) {
        const int top = 63 - (int)__builtin_clzll(NW);
        carry_after = ((MM >> top) & 1ull) != 0ull;
      }
      if (MM) carry_pc = __shfl(e0, 63 - (int)__builtin_clzll(MM)) - 1u;
      first_step = false;
    }
    if (chunk_bad) break;
    written += chunk_bytes;
    cs = ce;
    rk = rk_next;
  }
  if (!FILL && lane == 0u) {
    nbytes[k] = written;
    wga_vcf_err_dev e2;
    e2.item = bad_kind ? bad_item : WGA_NONE, e2.kind = bad_kind, e2.ch = bad_ch;
    err[k] = e2;
  }
}

#endif /* WGA_K19_MAF_CALL_H */
