/*
 * wga_k12_dotplot.h — K12: dotplot base-level segments (emit_baseplotdatas, cigar.rs:815-914).
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K12_DOTPLOT_H
#define WGA_K12_DOTPLOT_H

#include "wga_kernels.h"

/* ============================================================================================ */
/* K12: dotplot base-level segments (emit_baseplotdatas, cigar.rs:815-914) */
/* ============================================================================================ */
/* The reference folds the ops with a `last_m` flag: an I / D longer than the cutoff is a segment
 * of its own and closes the open M segment; an M-like op opens an M segment unless one is open;
 * every other M / small I / small D moves the end of the open segment to the running offsets
 * (small indels with no open segment only advance the offsets; ops outside M = X I D are
 * ignored).  Read as intervals between "breaks" (long indels): an interval holds one M segment
 * iff it has an M-like op; it starts at the offsets of the first such op and ends at the offsets
 * at the end of the interval.  So the opener writes the start fields and the closer — the next
 * break, or the end of the record — writes the end fields, and nothing is serial.
 * One wave per record, 256 ops per step.  Segment = 5 u64: ref_start, ref_end, query_start,
 * query_end (the two swapped for '-' records, cigar.rs:807-812), kind 0 M / 1 I / 2 D.
 * A record with a split (>= 2^28) I / D goes through the serial walk: its pieces count as one op. */
#define WGA_SEG_WORDS 5u
__device__ __forceinline__ void seg_write(u64* s, u64 rs, u64 re, u64 qs, u64 qe, u64 kind, bool neg) {
  s[0] = rs;
  s[1] = re;
  s[2] = neg ? qe : qs;
  s[3] = neg ? qs : qe;
  s[4] = kind;
}
__device__ __forceinline__ u64 dotplot_serial(const u32* rec, u64 nops, u64 cutoff, u64 r, u64 q, bool neg,
                                              u64* segs) {
  u64 ns = 0;
  bool last_m = false;
  for (u64 k = 0; k < nops;) {
    const u32 code = rec[k] & 15u;
    u64 len = rec[k] >> 4;
    u64 k2 = k + 1;
    if (code == WGA_OP_I || code == WGA_OP_D) /* pieces of one split length */
      while (k2 < nops && (rec[k2] & 15u) == (code == WGA_OP_I ? (u32)WGA_OP_I_CONT : (u32)WGA_OP_D_CONT)) len += rec[k2++] >> 4;
    const bool isi = code == WGA_OP_I || code == WGA_OP_I_CONT, isd = code == WGA_OP_D || code == WGA_OP_D_CONT;
    if (code == WGA_OP_M || code == WGA_OP_EQ || code == WGA_OP_X) {
      if (!last_m) {
        if (segs) seg_write(segs + ns * WGA_SEG_WORDS, r, r + len, q, q + len, 0, neg);
        ns++;
      } else if (segs) {
        u64* s = segs + (ns - 1) * WGA_SEG_WORDS;
        s[1] = r + len;
        s[neg ? 2 : 3] = q + len;
      }
      r += len;
      q += len;
      last_m = true;
    } else if (isi || isd) {
      const u64 re = isd ? r + len : r, qe = isi ? q + len : q;
      if (len > cutoff) {
        if (segs) seg_write(segs + ns * WGA_SEG_WORDS, r, re, q, qe, isi ? 1 : 2, neg);
        ns++;
        last_m = false;
      } else if (last_m && segs) {
        u64* s = segs + (ns - 1) * WGA_SEG_WORDS;
        if (isd) s[1] = re; else s[neg ? 2 : 3] = qe;
      }
      r = re;
      q = qe;
    }
    k = k2;
  }
  return ns;
}

/* The walk of ops [a, b) of one record by one wave (a a multiple of 256): offsets r / q, the number of segments so far and the
 * open / closed state in front of the range come in through `st` and leave through it; `first_ev` = the first event of the
 * range (0 none, 1 a break, 2 an M-like op) tells the caller whether a range that follows an open M segment starts a new
 * one.  Returns false when the range holds a continuation piece of a split indel (the record then takes the serial walk). */
struct DotState {
  u64 r, q, nseg;
  u32 state; /* 0 / 1: no open M segment (start, or a break was the last event), 2: open */
};
template <bool FILL>
__device__ __forceinline__ bool dotplot_walk(const u32* __restrict__ rec, u64 a, u64 b, u64 cutoff, bool neg, DotState& ds,
                                             u64* out, u32 lane, u32* first_ev) {
  u64 r_base = ds.r, q_base = ds.q, nseg = ds.nseg;
  u32 carry_state = ds.state, first_seen = 0u;
  u32 wnext[4]; /* the next step's ops travel behind the work on this step's */
#pragma unroll
  for (int e = 0; e < 4; e++) wnext[e] = a + (u64)lane * 4u + (u64)e < b ? rec[a + (u64)lane * 4u + e] : 0xFu;
  for (u64 k0 = a; k0 < b; k0 += 256) {
    const u64 kb = k0 + (u64)lane * 4u;
    u32 len[4], radv[4], qadv[4];
    bool ml[4], brk[4], isi[4];
    bool cont = false;
    u32 sr = 0, sq = 0, last_ev = 0, lane_first = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const u32 w = wnext[e];
      wnext[e] = kb + 256u + (u64)e < b ? rec[kb + 256u + e] : 0xFu;
      const u32 code = w & 15u;
      len[e] = w >> 4;
      ml[e] = code == WGA_OP_M || code == WGA_OP_EQ || code == WGA_OP_X;
      isi[e] = code == WGA_OP_I;
      const bool isd = code == WGA_OP_D;
      cont |= code == WGA_OP_I_CONT || code == WGA_OP_D_CONT;
      brk[e] = (isi[e] || isd) && (u64)len[e] > cutoff;
      radv[e] = (ml[e] || isd) ? len[e] : 0u;
      qadv[e] = (ml[e] || isi[e]) ? len[e] : 0u;
      sr += radv[e];
      sq += qadv[e];
      const u32 ev = ml[e] ? 2u : brk[e] ? 1u : 0u;
      last_ev = ev ? ev : last_ev;
      lane_first = lane_first ? lane_first : ev;
    }
    if (__ballot(cont)) return false; /* wave-uniform */
    /* offsets in front of this lane's ops: exact wave scans of the lane sums (< 2^30) on 16-bit halves */
    const u32 rl = wave_incl_scan_u32(sr & 0xFFFFu), rh = wave_incl_scan_u32(sr >> 16);
    const u32 ql = wave_incl_scan_u32(sq & 0xFFFFu), qh = wave_incl_scan_u32(sq >> 16);
    u64 r = r_base + (((u64)rh << 16) + (u64)rl) - (u64)sr;
    u64 q = q_base + (((u64)qh << 16) + (u64)ql) - (u64)sq;
    /* open / closed in front of this lane = the last event of the nearest earlier lane that has one */
    const u64 all_ev = __ballot(last_ev != 0u);
    const u64 evm = all_ev & ((1ull << lane) - 1ull);
    const int src = evm ? 63 - (int)__builtin_clzll(evm) : 0;
    const u32 got = (u32)__shfl((int)last_ev, src);
    u32 state = evm ? got : carry_state;
    if (!first_seen && all_ev) first_seen = (u32)__shfl((int)lane_first, (int)__builtin_ctzll(all_ev));
    /* segments this lane raises, then their ranks */
    u32 st = state, cnt = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      cnt += (brk[e] || (ml[e] && st != 2u)) ? 1u : 0u;
      st = ml[e] ? 2u : brk[e] ? 1u : st;
    }
    const u32 cinc = wave_incl_scan_u32(cnt);
    if (FILL) {
      u64 idx = nseg + (u64)(cinc - cnt);
      st = state;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (brk[e]) {
          if (st == 2u) { /* closes the open M segment */
            u64* s = out + (idx - 1) * WGA_SEG_WORDS;
            s[1] = r;
            s[neg ? 2 : 3] = q;
          }
          seg_write(out + idx * WGA_SEG_WORDS, r, r + radv[e], q, q + qadv[e], isi[e] ? 1 : 2, neg);
          idx++;
        } else if (ml[e] && st != 2u) { /* opens one: the end fields come from its closer */
          u64* s = out + idx * WGA_SEG_WORDS;
          s[0] = r;
          s[neg ? 3 : 2] = q;
          s[4] = 0;
          idx++;
        }
        st = ml[e] ? 2u : brk[e] ? 1u : st;
        r += radv[e];
        q += qadv[e];
      }
    }
    nseg += (u64)wave_last_u32(cinc);
    r_base += ((u64)wave_last_u32(rh) << 16) + (u64)wave_last_u32(rl);
    q_base += ((u64)wave_last_u32(qh) << 16) + (u64)wave_last_u32(ql);
    if (all_ev) carry_state = (u32)__shfl((int)last_ev, 63 - (int)__builtin_clzll(all_ev));
  }
  ds.r = r_base;
  ds.q = q_base;
  ds.nseg = nseg;
  ds.state = carry_state;
  *first_ev = first_seen;
  return true;
}

/* one wave per record; records beyond `long_ops` ops are left to the piece kernels below (long_ops = 0: none are) */
template <bool FILL>
__global__ __launch_bounds__(256) void k_dotplot_segments(u32 n, const u32* __restrict__ ops,
                                                          const u64* __restrict__ op_off,
                                                          const u8* __restrict__ strand_neg, u64 cutoff,
                                                          const u64* __restrict__ t_start,
                                                          const u64* __restrict__ q_start, u64* seg_cnt,
                                                          u64* segs, const u64* seg_off, u64 long_ops) {
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 i = (u64)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
  if (long_ops && nops > long_ops) return;
  const u32* rec = ops + o0;
  const bool neg = strand_neg[i] != 0;
  u64* const out = FILL ? segs + seg_off[i] * WGA_SEG_WORDS : (u64*)0;
  DotState ds;
  ds.r = t_start[i], ds.q = q_start[i], ds.nseg = 0, ds.state = 0u;
  u32 first_ev;
  if (!dotplot_walk<FILL>(rec, 0, nops, cutoff, neg, ds, out, lane, &first_ev)) { /* wave-uniform */
    if (lane == 0) {
      const u64 ns = dotplot_serial(rec, nops, cutoff, t_start[i], q_start[i], neg, out);
      if (!FILL) seg_cnt[i] = ns;
    }
    return;
  }
  if (lane == 0) {
    if (FILL && ds.state == 2u) { /* the end of the record closes the open M segment */
      u64* s = out + (ds.nseg - 1) * WGA_SEG_WORDS;
      s[1] = ds.r;
      s[neg ? 2 : 3] = ds.q;
    }
    if (!FILL) seg_cnt[i] = ds.nseg;
  }
}

/* ---- long records in pieces (see the piece kernels of K7): a first walk leaves each piece's offset sums, its segment count
 *      as if nothing were open in front of it, its first event and the state behind it; one thread per record turns that into
 *      each piece's start (offsets, segments so far, open / closed); the second walk writes the segments and the record's
 *      last piece closes the segment that is still open. ----------------------------------------------------------------- */
struct wga_dot_piece {
  u64 r, q, nseg; /* MODE 0: sums / count (start closed); after the record scan: the piece's start */
  u32 state;      /* MODE 0: state behind the piece, 0 when it has no event; after the scan: state in front of it */
  u32 first_ev;   /* MODE 0: 0 none, 1 break, 2 M-like, 3: a continuation piece of a split indel (serial walk); after the
                     scan: 3 = the record takes the serial walk */
};
template <int MODE>
__global__ __launch_bounds__(256) void k_dotplot_pieces(u32 n, const u32* __restrict__ ops, const u64* __restrict__ op_off,
                                                        const u8* __restrict__ strand_neg, u64 cutoff,
                                                        const u64* __restrict__ t_start, const u64* __restrict__ q_start,
                                                        const u64* __restrict__ piece_off, const u32* __restrict__ piece_rec, wga_dot_piece* pc,
                                                        u64* segs, const u64* seg_off) {
  const u32 lane = threadIdx.x & 63u;
  const u64 n_pieces = piece_off[n];
  const u64 n_waves = (u64)gridDim.x * 4u;
  for (u64 p = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x); p < n_pieces; p += n_waves) {
    const u32 i = WGA_UNI32(piece_rec[p]);
    const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
    const u64 psz = piece_span(nops, piece_off[i + 1] - piece_off[i]);
    const u64 a0 = (p - piece_off[i]) * psz, a = a0 < nops ? a0 : nops, b = a + psz < nops ? a + psz : nops;
    const bool neg = strand_neg[i] != 0;
    DotState ds;
    u32 first_ev = 0u;
    if (MODE == 0) {
      ds.r = ds.q = ds.nseg = 0;
      ds.state = 0u;
      const bool ok = dotplot_walk<false>(ops + o0, a, b, cutoff, neg, ds, (u64*)0, lane, &first_ev);
      if (lane == 0) {
        wga_dot_piece r;
        r.r = ds.r, r.q = ds.q, r.nseg = ds.nseg;
        r.state = first_ev ? ds.state : 0u;
        r.first_ev = ok ? first_ev : 3u;
        pc[p] = r;
      }
    } else {
      const wga_dot_piece r = pc[p];
      u64* const out = segs + seg_off[i] * WGA_SEG_WORDS;
      if (r.first_ev == 3u) { /* a split indel somewhere in the record: the serial walk, by the first piece's first lane */
        if (a == 0 && lane == 0) dotplot_serial(ops + o0, nops, cutoff, t_start[i], q_start[i], neg, out);
        continue;
      }
      ds.r = r.r, ds.q = r.q, ds.nseg = r.nseg, ds.state = r.state;
      dotplot_walk<true>(ops + o0, a, b, cutoff, neg, ds, out, lane, &first_ev);
      if (b == nops && lane == 0 && ds.state == 2u) { /* the end of the record closes the open M segment */
        u64* s = out + (ds.nseg - 1) * WGA_SEG_WORDS;
        s[1] = ds.r;
        s[neg ? 2 : 3] = ds.q;
      }
    }
  }
}
/* one thread per long record: piece sums -> piece starts; the record's segment count */
__global__ __launch_bounds__(256) void k_dotplot_piece_scan(u32 n, const u32* __restrict__ ops, const u64* __restrict__ op_off,
                                                            const u8* __restrict__ strand_neg, u64 cutoff,
                                                            const u64* __restrict__ t_start, const u64* __restrict__ q_start,
                                                            const u64* __restrict__ piece_off, wga_dot_piece* pc, u64* seg_cnt) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 p0 = piece_off[i], p1 = piece_off[i + 1];
  if (p0 == p1) return;
  bool weird = false;
  for (u64 p = p0; p < p1; p++) weird |= pc[p].first_ev == 3u;
  if (weird) {
    for (u64 p = p0; p < p1; p++) pc[p].first_ev = 3u;
    if (seg_cnt)
      seg_cnt[i] = dotplot_serial(ops + op_off[i], op_off[i + 1] - op_off[i], cutoff, t_start[i], q_start[i], strand_neg[i] != 0, (u64*)0);
    return;
  }
  u64 r = t_start[i], q = q_start[i], ns = 0;
  u32 state = 0u;
  for (u64 p = p0; p < p1; p++) {
    const wga_dot_piece x = pc[p];
    wga_dot_piece s;
    s.r = r, s.q = q, s.nseg = ns, s.state = state, s.first_ev = x.first_ev;
    pc[p] = s;
    r += x.r;
    q += x.q;
    /* counted as if nothing were open: an M-like first event continues the segment that is */
    ns += x.nseg - ((state == 2u && x.first_ev == 2u) ? 1u : 0u);
    state = x.first_ev ? x.state : state;
  }
  if (seg_cnt) seg_cnt[i] = ns;
}

#endif /* WGA_K12_DOTPLOT_H */
