/*
 * wga_k10_chain.h — K10: paf2chain data lines and header trims (cigar.rs:202-295,460-490).
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K10_CHAIN_H
#define WGA_K10_CHAIN_H

#include "wga_kernels.h"

/* ============================================================================================ */
/* K10: paf2chain data lines                                              */
/* ============================================================================================ */
/* parse_cigar_to_chain + cigar_unit_chain (cigar.rs:251-295,460-490): runs of M / = / X ops form a
 * block; when an M-like op follows an indel group and a block is open, the line
 * "\n<size>\t<D bases of the group>\t<I bases of the group>" goes out; leading indels are dropped,
 * the last block ends the record as "\n<size>" (trailing indels dropped).  parse_cigar_to_trim
 * (cigar.rs:202-245) for the chain header: I / D bases before the first M-like op, and the
 * length of the LAST I / D op behind the last M-like op (assignment, not a sum).
 * One wave per record, 4 consecutive ops per lane and 256 per step.  With exclusive prefix sums
 * PM, PD, PI of the M-like / D / I lengths, the line raised at op j is the difference between the
 * prefix triple at j and the triple at the previous raising op (the first M-like op for the first
 * line): the triples of a step go through LDS so that every raising op can read its predecessor.
 * Two passes: text bytes per record, then the text. */
struct wga_chain_trim {
  u64 head_ins, head_del, tail_ins, tail_del;
};

/* the reference's loops verbatim on packed ops (lane 0 only): used for records with zero-length ops,
 * whose "size != 0" / "diffs != 0" tests (cigar.rs:472) the prefix formulation does not cover */
__device__ __forceinline__ void chain_serial(const u32* rec, u64 nops, u8* text, wga_chain_trim& tr,
                                             u64& nbytes, u64& bad_idx) {
  u64 size = 0, qd = 0, td = 0, off = 0;
  u64 head_ins = 0, head_del = 0, tail_ins = 0, tail_del = 0;
  bool head = true;
  bad_idx = WGA_NONE;
  for (u64 k = 0; k < nops; k++) {
    const u32 code = rec[k] & 15u;
    const u64 len = rec[k] >> 4;
    if (code == WGA_OP_M || code == WGA_OP_EQ || code == WGA_OP_X) {
      if (size != 0 && td + qd != 0) {
        const u32 a = dec_digits(size), b = dec_digits(qd), c = dec_digits(td);
        if (text) {
          u8* p = text + off;
          *p++ = (u8)'\n';
          dec_write(p, size, a);
          p += a;
          *p++ = (u8)'\t';
          dec_write(p, qd, b);
          p += b;
          *p++ = (u8)'\t';
          dec_write(p, td, c);
        }
        off += 3u + a + b + c;
        size = 0;
      }
      size += len;
      td = qd = 0;
      tail_ins = tail_del = 0;
      head = false;
    } else if (code == WGA_OP_I || code == WGA_OP_I_CONT) {
      td += len;
      if (head) head_ins += len;
      tail_ins = code == WGA_OP_I ? len : tail_ins + len;
    } else if (code == WGA_OP_D || code == WGA_OP_D_CONT) {
      qd += len;
      if (head) head_del += len;
      tail_del = code == WGA_OP_D ? len : tail_del + len;
    } else {
      bad_idx = k;
      break;
    }
  }
  const u32 dl = dec_digits(size);
  if (text && bad_idx == WGA_NONE) {
    u8* p = text + off;
    *p++ = (u8)'\n';
    dec_write(p, size, dl);
  }
  nbytes = off + 1u + dl;
  tr.head_ins = head_ins;
  tr.head_del = head_del;
  tr.tail_ins = tail_ins;
  tr.tail_del = tail_del;
}

/* ---- the wave-parallel walk ------------------------------------------------------------------
 * A step is WGA_CHAIN_STEP = 64 x WGA_CHAIN_OPL consecutive ops, WGA_CHAIN_OPL consecutive ops per lane, kept in
 * registers.  An op RAISES a line when it is M-like and the op in front of it is not (the record's first op never
 * raises).  Per lane, one unrolled pass over its ops runs the reference's accumulators (size, D bases, I bases since
 * the last raise) and stores their value at every raise into the wave's slot list in LDS, at the rank a wave scan of
 * the raise counts gives; what is open at the lane's end goes through three wave scans so that every lane learns what
 * was open at its start (from the nearest lower lane that raised, or from the previous step), and adds that to its
 * first line.  The lines are then formatted DENSELY, one line per lane and 64 per round — digit counts in the count
 * pass; in the fill pass the text of a round is put together in an LDS buffer that mirrors the output's position
 * inside its 128-byte line, and only whole lines are stored (16 bytes per lane; the rest waits for the next round):
 * no line of the output is written in pieces.
 * The state between steps is the reference's own (size, D bases, I bases of its loop, cigar.rs:460-490), and the
 * reference's tests are on VALUES ("size != 0", "diffs != 0", cigar.rs:472): with lengths >= 1 they say "M-like op behind
 * an indel op, and not the record's first M-like op", which is what the lane pass finds by op class (a line whose size
 * is 0 — the first M-like op behind leading indels — is not printed; its D / I sums are the head trim).  A step that
 * holds a zero-length op or an op outside M = X I D (a wave vote) runs the reference's loop as it stands over its ops —
 * scalar, the words read out of the lanes' registers — and hands its lines to the same formatting rounds; the fold
 * ends at the op outside M = X I D.  Everything is u32 and exact as long as the open sums plus a step's lengths stay
 * below 2^32 (voted per step); a record that passes that goes through chain_serial as a whole.  The tail trim is
 * read off the record's last ops afterwards. */
#ifndef WGA_CHAIN_OPL
#define WGA_CHAIN_OPL 8
#endif
#define WGA_CHAIN_STEP (64u * WGA_CHAIN_OPL)
#define WGA_CHAIN_NL (32u * WGA_CHAIN_OPL) /* at most every other op raises */
#define WGA_CHAIN_TB 2432u                 /* < 128 carried bytes + 64 lines x 33 bytes + the last line, + 128 read ahead */
#define WGA_CHAIN_MM ((1u << WGA_OP_M) | (1u << WGA_OP_EQ) | (1u << WGA_OP_X))
#define WGA_CHAIN_IM ((1u << WGA_OP_I) | (1u << WGA_OP_I_CONT))
#define WGA_CHAIN_DM ((1u << WGA_OP_D) | (1u << WGA_OP_D_CONT))
#define WGA_CHAIN_PAD 0xFu /* behind the record's end: no length, no class */

/* decimal digits of a u32; p10[t] = 10^t, t < 10 */
__device__ __forceinline__ u32 dec_digits_u32(u32 v, const u32* p10) {
  const u32 x = v | 1u;
  const u32 t = ((32u - (u32)__clz(x)) * 1233u) >> 12;
  return t + 1u - (x < p10[t] ? 1u : 0u);
}

struct ChainWalk {
  u32 c_size, c_qd, c_td; /* the reference's size / query_diff / target_diff in front of the step (wave-uniform) */
  u32 seen_m;             /* an M-like op was seen: the head trim is closed */
  u32 head_ins, head_del;
  u64 nbytes;             /* count pass: this lane's share of the text bytes */
  u32 fill, head_skip;    /* fill pass: bytes in the text buffer; those in front of head_skip are not this record's */
  u8* gpos;               /* fill pass: where byte 0 of the text buffer belongs (128-byte aligned) */
};

__device__ __forceinline__ void chain_load(const u32* rec, u64 nops, u64 k0, u32 lane, u32 (&w)[WGA_CHAIN_OPL]) {
  const u64 kb = k0 + (u64)lane * WGA_CHAIN_OPL;
  if (kb + WGA_CHAIN_OPL <= nops) {
#pragma unroll
    for (int j = 0; j < (int)WGA_CHAIN_OPL / 4; j++) {
      const u32x4_a4 v = *(const u32x4_a4*)(rec + kb + 4 * j);
      w[4 * j] = v[0], w[4 * j + 1] = v[1], w[4 * j + 2] = v[2], w[4 * j + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) w[e] = kb + (u64)e < nops ? rec[kb + e] : WGA_CHAIN_PAD;
  }
}

/* whole 128-byte lines of the text buffer go out, the rest moves to the buffer's start */
__device__ __forceinline__ void chain_flush_lines(ChainWalk& st, u8* tbuf, u32 lane) {
  const u32 nfull = st.fill >> 7;
  if (nfull == 0u) return; /* wave-uniform */
  WGA_WAVE_SYNC();
  for (u32 g = lane; g < nfull * 8u; g += 64u) {
    const u32 bo = g * 16u;
    if (bo >= st.head_skip) {
      *(u32x4_a16*)(st.gpos + bo) = *(const u32x4_a16*)(tbuf + bo);
    } else if (bo + 16u > st.head_skip) {
      for (u32 k = st.head_skip; k < bo + 16u; k++) st.gpos[k] = tbuf[k];
    }
  }
  const u32 keep = lane < 32u ? *(const u32*)(tbuf + nfull * 128u + lane * 4u) : 0u;
  WGA_WAVE_SYNC();
  if (lane < 32u) *(u32*)(tbuf + lane * 4u) = keep;
  WGA_WAVE_SYNC();
  st.gpos += (u64)nfull * 128u;
  st.fill &= 127u;
  st.head_skip = 0u;
}

/* v's decimal digits end in front of tbuf[end]; returns where they start */
__device__ __forceinline__ u32 chain_put_dec(u8* tbuf, u32 end, u32 v) {
  do {
    const u32 q = v / 10u;
    tbuf[--end] = (u8)('0' + (v - q * 10u));
    v = q;
  } while (v);
  return end;
}

/* the lines of a step by op class, 8 ops per lane: returns their number, leaves them in slot[] */
__device__ __forceinline__ u32 chain_find_lines(const u32 (&w)[WGA_CHAIN_OPL], u32 mb, u32 lane, u32x4_a16* slot,
                                                ChainWalk& st) {
  u32 pm = (u32)__shfl_up((int)(mb >> (WGA_CHAIN_OPL - 1u)), 1u);
  if (lane == 0u) pm = (st.c_qd | st.c_td) == 0u ? 1u : 0u; /* no indel open: the op in front is M-like, or none */
  const u32 rb = mb & ~((mb << 1) | (pm & 1u));
  const u32 cnt = (u32)__popc(rb);
  const u32 einc = wave_incl_scan_u32(cnt);
  const u32 nl = WGA_UNI32(wave_last_u32(einc));
  /* the accumulators; a raise stores them and starts them again */
  u32 pos = einc - cnt, size = 0, qd = 0, td = 0;
#pragma unroll
  for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) {
    const u32 code = w[e] & 15u, len = w[e] >> 4;
    const u32 r = bit_mask(rb, (u32)e);
    if (r) {
      u32x4_a16 v;
      v[0] = size, v[1] = qd, v[2] = td, v[3] = 0u;
      slot[pos] = v;
    }
    pos -= r;
    size = (size & ~r) + (len & bit_mask(WGA_CHAIN_MM, code));
    qd = (qd & ~r) + (len & bit_mask(WGA_CHAIN_DM, code));
    td = (td & ~r) + (len & bit_mask(WGA_CHAIN_IM, code));
  }
  /* what was open at the lane's start */
  const u32 s0 = wave_incl_scan_u32(size), s1 = wave_incl_scan_u32(qd), s2 = wave_incl_scan_u32(td);
  const u32 x0 = s0 - size, x1 = s1 - qd, x2 = s2 - td;
  const u64 hb = __ballot(cnt != 0u);
  const u64 below = hb & ((1ull << lane) - 1ull);
  const int src = below ? 63 - (int)__builtin_clzll(below) : 0;
  const u32 y0 = (u32)__shfl((int)x0, src), y1 = (u32)__shfl((int)x1, src), y2 = (u32)__shfl((int)x2, src);
  if (cnt) {
    u32x4_a16 v = slot[einc - cnt];
    v[0] += below ? x0 - y0 : st.c_size + x0;
    v[1] += below ? x1 - y1 : st.c_qd + x1;
    v[2] += below ? x2 - y2 : st.c_td + x2;
    slot[einc - cnt] = v;
  }
  const int last = hb ? 63 - (int)__builtin_clzll(hb) : 0;
  const u32 z0 = (u32)__shfl((int)x0, last), z1 = (u32)__shfl((int)x1, last), z2 = (u32)__shfl((int)x2, last);
  const u32 t0 = wave_last_u32(s0), t1 = wave_last_u32(s1), t2 = wave_last_u32(s2);
  st.c_size = WGA_UNI32(hb ? t0 - z0 : st.c_size + t0);
  st.c_qd = WGA_UNI32(hb ? t1 - z1 : st.c_qd + t1);
  st.c_td = WGA_UNI32(hb ? t2 - z2 : st.c_td + t2);
  return nl;
}

/* the same by the reference's loop (cigar.rs:460-490 + the head of 202-245), one op after the other with wave-uniform
 * values: for steps that hold zero-length ops or ops outside M = X I D.  nops_step: the ops of the step that exist.
 * Returns the number of lines; *stop = index (inside the step) of the first op outside M = X I D, or ~0. */
__device__ __forceinline__ u32 chain_find_lines_serial(const u32 (&w)[WGA_CHAIN_OPL], u32 nops_step, u32 lane,
                                                       u32x4_a16* slot, ChainWalk& st, u32* stop) {
  u32 size = st.c_size, qd = st.c_qd, td = st.c_td, nl = 0;
  *stop = 0xFFFFFFFFu;
  for (u32 l = 0; l * WGA_CHAIN_OPL < nops_step; l++) {
    bool out = false;
#pragma unroll
    for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) {
      const u32 x = l * WGA_CHAIN_OPL + (u32)e;
      const u32 word = WGA_UNI32((u32)__shfl((int)w[e], (int)l));
      if (out || x >= nops_step) continue;
      const u32 code = word & 15u, len = word >> 4;
      if ((WGA_CHAIN_MM >> code) & 1u) {
        if (size != 0u && (qd | td) != 0u) {
          if (lane == 0u) {
            u32x4_a16 v;
            v[0] = size, v[1] = qd, v[2] = td, v[3] = 0u;
            slot[nl] = v;
          }
          nl++;
          size = 0u;
        }
        if (!st.seen_m) st.head_del = qd, st.head_ins = td, st.seen_m = 1u;
        size += len;
        qd = td = 0u;
      } else if ((WGA_CHAIN_IM >> code) & 1u) {
        td += len;
      } else if ((WGA_CHAIN_DM >> code) & 1u) {
        qd += len;
      } else {
        *stop = x;
        out = true;
      }
    }
    if (out) break;
  }
  st.c_size = size, st.c_qd = qd, st.c_td = td;
  return nl;
}

/* one step: 0 = go on, 1 = the fold ended at an op outside M = X I D (*stop), 2 = sums past 2^32: chain_serial */
template <bool FILL, bool TAIL>
__device__ __forceinline__ int chain_step(const u32 (&w)[WGA_CHAIN_OPL], u32 nops_step, u32 lane, u32x4_a16* slot,
                                          const u32* p10, u8* tbuf, ChainWalk& st, u32* stop) {
  /* classes as bit masks, the checks, the lane's total length */
  const u32 lo = lane * WGA_CHAIN_OPL;
  u32 mb = 0, seen = 0, minw = 0xFFFFFFFFu, tl = 0;
#pragma unroll
  for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) {
    const u32 wc = TAIL ? (lo + (u32)e < nops_step ? w[e] : (1u << 4 | WGA_OP_EQ)) : w[e];
    mb |= ((WGA_CHAIN_MM >> (w[e] & 15u)) & 1u) << e;
    seen |= 1u << (wc & 15u);
    minw = wc < minw ? wc : minw;
    tl += w[e] >> 4;
  }
  if ((u64)st.c_size + (u64)st.c_qd + (u64)st.c_td + wave_sum_u32_wide(tl) > 0xFFFFFFFFull) return 2;
  const bool odd = (seen & ~(WGA_CHAIN_MM | WGA_CHAIN_IM | WGA_CHAIN_DM)) != 0u || minw < 16u;
  const bool head_open = st.seen_m == 0u;
  u32 nl;
  *stop = 0xFFFFFFFFu;
  if (__ballot(odd)) { /* wave-uniform */
    nl = chain_find_lines_serial(w, nops_step, lane, slot, st, stop);
  } else {
    nl = chain_find_lines(w, mb, lane, slot, st);
    if (head_open && __ballot(mb != 0u)) st.seen_m = 1u; /* the head trim: line 0's sums when its size is 0, else 0 */
  }
  if (nl) { /* wave-uniform */
    WGA_WAVE_SYNC();
    /* the lines, 64 per round; one whose size is 0 is the first M-like op behind leading indels: not printed */
    for (u32 base = 0; base < nl; base += 64u) {
      const u32 j = base + lane;
      u32x4_a16 v;
      v[0] = v[1] = v[2] = v[3] = 0u;
      if (j < nl) v = slot[j];
      if (head_open && base == 0u) {
        const u32 hs = (u32)__shfl((int)v[0], 0), hd = (u32)__shfl((int)v[1], 0), hi = (u32)__shfl((int)v[2], 0);
        if (hs == 0u) st.head_del = hd, st.head_ins = hi;
      }
      const bool on = j < nl && v[0] != 0u;
      const u32 d0 = dec_digits_u32(v[0], p10), d1 = dec_digits_u32(v[1], p10), d2 = dec_digits_u32(v[2], p10);
      const u32 ll = on ? 3u + d0 + d1 + d2 : 0u;
      if (!FILL) {
        st.nbytes += ll;
      } else {
        const u32 linc = wave_incl_scan_u32(ll);
        if (on) {
          u32 p = st.fill + linc; /* the line's end */
          p = chain_put_dec(tbuf, p, v[2]);
          tbuf[--p] = (u8)'\t';
          p = chain_put_dec(tbuf, p, v[1]);
          tbuf[--p] = (u8)'\t';
          p = chain_put_dec(tbuf, p, v[0]);
          tbuf[--p] = (u8)'\n';
        }
        st.fill += WGA_UNI32(wave_last_u32(linc));
        chain_flush_lines(st, tbuf, lane);
      }
    }
    WGA_WAVE_SYNC(); /* the slots are rewritten by the next step */
  }
  return *stop != 0xFFFFFFFFu ? 1 : 0;
}

/* parse_cigar_to_trim's tail (cigar.rs:202-245): the length of the last I (D) op behind the last M-like op, a split
 * length being its head op plus the continuation pieces — the I-class (D-class) lengths from the last op that is
 * M-like or an I (D) head on.  Clean records only; read backwards, 64 ops per look. */
__device__ __forceinline__ void chain_tail_trim(const u32* rec, u64 nops, u32 lane, u64& tail_ins, u64& tail_del) {
  tail_ins = tail_del = 0;
  bool done_i = false, done_d = false;
  for (u64 hi = nops; hi > 0 && !(done_i && done_d);) {
    const u64 lo = hi > 64u ? hi - 64u : 0u;
    const bool in = lo + lane < hi;
    const u32 w = in ? rec[lo + lane] : WGA_CHAIN_PAD;
    const u32 code = w & 15u, len = w >> 4;
    const bool m = ((WGA_CHAIN_MM >> code) & 1u) != 0u;
    const bool ic = ((WGA_CHAIN_IM >> code) & 1u) != 0u, dc = ((WGA_CHAIN_DM >> code) & 1u) != 0u;
    if (!done_i) {
      const u64 stop = __ballot(m || code == WGA_OP_I);
      const u32 from = stop ? 63u - (u32)__builtin_clzll(stop) : 0u;
      tail_ins += wave_sum_u64(ic && lane >= from ? (u64)len : 0ull);
      done_i = stop != 0ull;
    }
    if (!done_d) {
      const u64 stop = __ballot(m || code == WGA_OP_D);
      const u32 from = stop ? 63u - (u32)__builtin_clzll(stop) : 0u;
      tail_del += wave_sum_u64(dc && lane >= from ? (u64)len : 0ull);
      done_d = stop != 0ull;
    }
    hi = lo;
  }
}

/* One wave walks `nops` ops from `rec` as a record of its own: `first` — the head trim is open (the record starts here);
 * `last` — the walk ends the record ("\n<size>", trailing indels dropped), otherwise an M-like op follows an open block
 * and an open indel group (a cut of chain_find_cut) and what is open goes out as a line.  Count pass: returns the text bytes;
 * fill pass: writes them at `text`.  *weird: sums past 2^32 (nothing usable was produced); *bad_idx: first op outside
 * M = X I D, relative to rec. */
template <bool FILL>
__device__ __forceinline__ u64 chain_walk(const u32* rec, u64 nops, u8* text, bool first, bool last, u32 lane,
                                          u32x4_a16* slot, const u32* p10, u8* tbuf, ChainWalk& st, bool* weird,
                                          u64* bad_idx) {
  st.c_size = st.c_qd = st.c_td = 0u;
  st.seen_m = first ? 0u : 1u;
  st.head_ins = st.head_del = 0u;
  st.nbytes = 0;
  st.fill = st.head_skip = FILL ? (u32)((uintptr_t)text & 127u) : 0u;
  st.gpos = FILL ? text - st.fill : (u8*)0;
  *weird = false;
  *bad_idx = WGA_NONE;
  u32 w[WGA_CHAIN_OPL];
  if (nops) chain_load(rec, nops, 0, lane, w);
  for (u64 k0 = 0; k0 < nops; k0 += WGA_CHAIN_STEP) {
    u32 nw[WGA_CHAIN_OPL];
    const bool more = k0 + WGA_CHAIN_STEP < nops;
    if (more) chain_load(rec, nops, k0 + WGA_CHAIN_STEP, lane, nw);
    u32 stop;
    const int rc = more || k0 + WGA_CHAIN_STEP == nops
                       ? chain_step<FILL, false>(w, WGA_CHAIN_STEP, lane, slot, p10, tbuf, st, &stop)
                       : chain_step<FILL, true>(w, (u32)(nops - k0), lane, slot, p10, tbuf, st, &stop);
    if (rc == 2) { /* wave-uniform */
      *weird = true;
      return 0;
    }
    if (rc == 1) {
      *bad_idx = k0 + (u64)stop;
      break;
    }
    if (more) {
#pragma unroll
      for (int e = 0; e < (int)WGA_CHAIN_OPL; e++) w[e] = nw[e];
    }
  }
  /* the last block: "\n<size>" (cigar.rs:289-291; 0 when the record has no M-like op); at a cut: the line the next op raises */
  const u32 d0 = dec_digits_u32(st.c_size, p10);
  const u32 el = last ? 1u + d0 : 3u + d0 + dec_digits_u32(st.c_qd, p10) + dec_digits_u32(st.c_td, p10);
  if (FILL) {
    if (lane == 0) {
      u32 p = st.fill + el;
      if (!last) {
        p = chain_put_dec(tbuf, p, st.c_td);
        tbuf[--p] = (u8)'\t';
        p = chain_put_dec(tbuf, p, st.c_qd);
        tbuf[--p] = (u8)'\t';
      }
      p = chain_put_dec(tbuf, p, st.c_size);
      tbuf[p - 1u] = (u8)'\n';
    }
    st.fill += el;
    chain_flush_lines(st, tbuf, lane);
    WGA_WAVE_SYNC();
    for (u32 k = st.head_skip + lane; k < st.fill; k += 64u) st.gpos[k] = tbuf[k];
    return 0;
  }
  return wave_sum_u64(st.nbytes) + el;
}

/* one wave per record; records beyond `long_ops` ops are left to the piece kernels below (long_ops = 0: none are) */
template <bool FILL>
__global__ __launch_bounds__(256) void k_cigar_chain(u32 n, const u32* __restrict__ ops,
                                                     const u64* __restrict__ op_off,
                                                     wga_chain_trim* trims, u64* nbytes,
                                                     wga_rec_diag* diag, u8* out,
                                                     const u64* out_off, u64 long_ops) {
  __shared__ u32x4_a16 s_slot[4][WGA_CHAIN_NL];
  __shared__ u32 s_p10[4][16];
  __shared__ u32x4_a16 s_text[4][FILL ? WGA_CHAIN_TB / 16u : 1u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 i = (u64)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
  if (long_ops && nops > long_ops) return;
  const u32* rec = ops + o0;
  u8* const text = FILL ? out + out_off[i] : (u8*)0;
  u32x4_a16* const slot = s_slot[wave];
  u32* const p10 = s_p10[wave];
  u8* const tbuf = (u8*)s_text[wave];
  if (lane < 10u) {
    u32 v = 1u;
    for (u32 k = 0; k < lane; k++) v *= 10u;
    p10[lane] = v;
  }
  WGA_WAVE_SYNC();
  ChainWalk st;
  bool weird;
  u64 bad_idx;
  const u64 nb = chain_walk<FILL>(rec, nops, text, true, true, lane, slot, p10, tbuf, st, &weird, &bad_idx);
  if (weird) { /* a block, or an indel group, of 2^32 bases or more: the reference's loop as it stands, in u64 */
    if (lane == 0) {
      wga_chain_trim tr;
      u64 nbs = 0, bad = WGA_NONE;
      chain_serial(rec, nops, text, tr, nbs, bad);
      if (!FILL) {
        nbytes[i] = nbs;
        trims[i] = tr;
        if (bad != WGA_NONE) diag[i].bad_op_idx = bad;
      }
    }
    return;
  }
  if (!FILL) {
    wga_chain_trim tr;
    /* indels in front of the first M-like op; everything when no M-like op exists */
    tr.head_ins = st.seen_m ? (u64)st.head_ins : (u64)st.c_td;
    tr.head_del = st.seen_m ? (u64)st.head_del : (u64)st.c_qd;
    chain_tail_trim(rec, nops, lane, tr.tail_ins, tr.tail_del);
    if (lane == 0) {
      nbytes[i] = nb;
      trims[i] = tr;
      if (bad_idx != WGA_NONE) diag[i].bad_op_idx = bad_idx;
    }
  }
}

/* ---- long records in pieces: the fold's state is (size, D bases, I bases) and it starts again at every line, so a record
 *      can be cut where a line is certain: op k is M-like, op k-1 an indel op of length >= 1, op k-2 an M-like op of length
 *      >= 1 ("size != 0 && diffs != 0" holds at k whatever came before, cigar.rs:472).  Piece j of a record beyond
 *      `long_ops` ops starts at the first such k at or behind j * piece_span (none inside its span: the piece is
 *      empty and its ops stay with the piece in front) and ends where the next non-empty piece starts; it is walked as a
 *      record of its own (chain_walk) whose last line is the one op k raises.  Count walk: bytes per piece; one thread per
 *      record: the record's bytes and every piece's place in its text; fill walk: the text.  A piece whose sums pass 2^32
 *      sends its record through chain_serial (by the scan thread, then by piece 0's lane 0). ------------------------------ */
struct wga_chain_piece {
  u64 a, b;   /* ops [a, b) of the record; a = WGA_NONE: empty */
  u64 nb;     /* count walk: text bytes (WGA_NONE: sums past 2^32) */
  u64 off;    /* record scan: the piece's text starts here inside the record's (WGA_NONE: the record is chain_serial's) */
};
__device__ __forceinline__ u64 chain_find_cut(const u32* rec, u64 lo, u64 hi, u32 lane) {
  for (u64 base = lo < 2u ? 2u : lo; base < hi; base += 64u) {
    const u64 k = base + lane;
    bool ok = false;
    if (k < hi) {
      const u32 w0 = rec[k], w1 = rec[k - 1], w2 = rec[k - 2];
      ok = ((WGA_CHAIN_MM >> (w0 & 15u)) & 1u) && (((WGA_CHAIN_IM | WGA_CHAIN_DM) >> (w1 & 15u)) & 1u) && (w1 >> 4) != 0u &&
           ((WGA_CHAIN_MM >> (w2 & 15u)) & 1u) && (w2 >> 4) != 0u;
    }
    const u64 m = __ballot(ok);
    if (m) return base + (u64)(__ffsll((unsigned long long)m) - 1);
  }
  return WGA_NONE;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_cigar_chain_pieces(u32 n, const u32* __restrict__ ops, const u64* __restrict__ op_off,
                                                            const u64* __restrict__ piece_off, const u32* __restrict__ piece_rec,
                                                            wga_chain_piece* pc, wga_chain_trim* trims, wga_rec_diag* diag,
                                                            u8* out, const u64* out_off) {
  __shared__ u32x4_a16 s_slot[4][WGA_CHAIN_NL];
  __shared__ u32 s_p10[4][16];
  __shared__ u32x4_a16 s_text[4][MODE ? WGA_CHAIN_TB / 16u : 1u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  u32x4_a16* const slot = s_slot[wave];
  u32* const p10 = s_p10[wave];
  u8* const tbuf = (u8*)s_text[wave];
  if (lane < 10u) {
    u32 v = 1u;
    for (u32 k = 0; k < lane; k++) v *= 10u;
    p10[lane] = v;
  }
  WGA_WAVE_SYNC();
  const u64 n_pieces = piece_off[n];
  const u64 n_waves = (u64)gridDim.x * 4u;
  for (u64 p = (u64)blockIdx.x * 4 + wave; p < n_pieces; p += n_waves) {
    const u32 i = WGA_UNI32(piece_rec[p]);
    const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
    const u32* rec = ops + o0;
    const u64 j = p - piece_off[i], np = piece_off[i + 1] - piece_off[i];
    ChainWalk st;
    bool weird;
    u64 bad;
    if (MODE == 0) {
      wga_chain_piece r;
      r.nb = 0, r.off = 0;
      const u64 psz = piece_span(nops, np);
      const u64 lo = j * psz;
      r.a = j == 0 ? 0 : chain_find_cut(rec, lo, lo + psz < nops ? lo + psz : nops, lane);
      r.b = WGA_NONE;
      if (r.a != WGA_NONE) {
        for (u64 jj = j + 1; jj < np && r.b == WGA_NONE; jj++) {
          const u64 l2 = jj * psz;
          r.b = chain_find_cut(rec, l2, l2 + psz < nops ? l2 + psz : nops, lane);
        }
        if (r.b == WGA_NONE) r.b = nops;
        const u64 nb = chain_walk<false>(rec + r.a, r.b - r.a, (u8*)0, j == 0, r.b == nops, lane, slot, p10, tbuf, st, &weird,
                                         &bad);
        r.nb = weird ? WGA_NONE : nb;
        if (bad != WGA_NONE && lane == 0 && diag) atomicMin((u64*)&diag[i].bad_op_idx, r.a + bad);
        if (j == 0 && !weird && trims) { /* wave-uniform */
          wga_chain_trim tr;
          tr.head_ins = st.seen_m ? (u64)st.head_ins : (u64)st.c_td;
          tr.head_del = st.seen_m ? (u64)st.head_del : (u64)st.c_qd;
          chain_tail_trim(rec, nops, lane, tr.tail_ins, tr.tail_del);
          if (lane == 0) trims[i] = tr;
        }
      }
      if (lane == 0) pc[p] = r;
    } else {
      const wga_chain_piece r = pc[p];
      u8* const text = out + out_off[i];
      if (r.off == WGA_NONE) {
        if (j == 0 && lane == 0) {
          wga_chain_trim tr;
          u64 nbs = 0, bd = WGA_NONE;
          chain_serial(rec, nops, text, tr, nbs, bd);
        }
      } else if (r.a != WGA_NONE) {
        chain_walk<true>(rec + r.a, r.b - r.a, text + r.off, j == 0, r.b == nops, lane, slot, p10, tbuf, st, &weird, &bad);
      }
    }
    WGA_WAVE_SYNC(); /* the next piece reuses the wave's slots and text buffer */
  }
}
/* one thread per long record: the record's bytes, every piece's place in the text */
__global__ __launch_bounds__(256) void k_cigar_chain_piece_scan(u32 n, const u32* __restrict__ ops,
                                                                const u64* __restrict__ op_off,
                                                                const u64* __restrict__ piece_off, wga_chain_piece* pc,
                                                                wga_chain_trim* trims, u64* nbytes, wga_rec_diag* diag) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 p0 = piece_off[i], p1 = piece_off[i + 1];
  if (p0 == p1) return;
  u64 sum = 0;
  bool weird = false;
  for (u64 p = p0; p < p1; p++) {
    const u64 nb = pc[p].nb;
    weird = weird || nb == WGA_NONE;
    pc[p].off = sum;
    sum += nb;
  }
  if (weird) {
    const u64 o0 = op_off[i];
    wga_chain_trim tr;
    u64 bad = WGA_NONE;
    chain_serial(ops + o0, op_off[i + 1] - o0, (u8*)0, tr, sum, bad);
    if (nbytes) trims[i] = tr, diag[i].bad_op_idx = bad;
    for (u64 p = p0; p < p1; p++) pc[p].off = WGA_NONE;
  }
  if (nbytes) nbytes[i] = sum;
}

#endif /* WGA_K10_CHAIN_H */
