/*
 * wga_k7_paf_call.h — K7: the op walk of call on PAF (call_within_var_paf, caller.rs:610-822) and the piece table of the op walks.
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K7_PAF_CALL_H
#define WGA_K7_PAF_CALL_H

#include "wga_kernels.h"

/* ============================================================================================ */
/* K7: PAF call op walk                                                                         */
/* ============================================================================================ */
/* call_within_var_paf (caller.rs:610-822) walks the op stream with two running positions and an
 * `after_m` flag and raises events at X ops (when SNPs are asked for) and at I / D ops longer
 * than the cutoff that directly follow an M / = / X op.  Per record one wave scans the ops 64 at
 * a time: exclusive u64 prefix sums of the target / query advance, `after_m` from the previous
 * op's code, compaction of the event ops by ballot.  The walk stops at the first op that is not
 * M = X I D (the reference's fold keeps its Err and skips the rest, :673,815-819).
 * A length >= 2^28 is packed as a head op plus continuation pieces (codes 9 / 10): the head is
 * flagged when `len > svlen` or a continuation follows; the host applies the cutoff to the sum.
 * Event entry = 3 u64: op index in the record, target advance before it, query advance before it. */

/* The walk of ops [a, b) of one record by one wave, 4 consecutive ops per lane and 256 per step (a is a multiple of 256): running
 * target / query positions and event count start from `st`, the op in front of a gives `after_m`, the op behind a step
 * tells whether an indel goes on in a continuation piece.  Events go to eout + 3 * (running count) when eout is given.
 * Returns the sums over the walked ops in *tot (events counted up to the walk's stop) and the record-relative index of
 * the first op outside M = X I D in *bad_at (WGA_NONE: none) — the walk stops there, as the reference's fold does before
 * it discards the error (caller.rs:673,815-819). */
struct PafCallState {
  u64 t, q, e;
};
/* POS = false: only the sums are wanted (a count pass): no prefix scans, every lane keeps its own sums and the wave adds them
 * up once when the walk ends */
template <bool POS>
__device__ __forceinline__ void paf_call_walk(const u32* __restrict__ rec, u64 nops, u64 a, u64 b, u64 svlen, u32 snp,
                                              PafCallState st, u64* eout, u32 lane, PafCallState* tot, u64* bad_at) {
  u64 t_base = st.t, q_base = st.q, e_base = st.e;
  u64 acc_t = 0, acc_q = 0; /* !POS: this lane's target / query advance ... */
  u32 acc_e = 0;            /* ... and events */
  u32 carry_code = a ? (rec[a - 1] & 15u) : 0xFu;
  *bad_at = WGA_NONE;
  u32 wnext[4]; /* the next step's ops travel behind the work on this step's */
#pragma unroll
  for (int e = 0; e < 4; e++) wnext[e] = a + (u64)lane * 4u + (u64)e < b ? rec[a + (u64)lane * 4u + e] : 0xFu; /* 0xF: no op */
  for (u64 k0 = a; k0 < b; k0 += 256) {
    const u64 kb = k0 + (u64)lane * 4u;
    u32 w[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      w[e] = wnext[e];
      wnext[e] = kb + 256u + (u64)e < b ? rec[kb + 256u + e] : 0xFu;
    }
    u32 code[4], len[4];
    bool valid[4], isi[4], isd[4];
    u32 firstbad = 4u; /* first op of this lane outside M = X I D (and inside the range) */
#pragma unroll
    for (int e = 3; e >= 0; e--) {
      code[e] = w[e] & 15u;
      len[e] = w[e] >> 4;
      const bool mlike = code[e] == WGA_OP_M || code[e] == WGA_OP_EQ || code[e] == WGA_OP_X;
      isi[e] = code[e] == WGA_OP_I || code[e] == WGA_OP_I_CONT;
      isd[e] = code[e] == WGA_OP_D || code[e] == WGA_OP_D_CONT;
      valid[e] = mlike || isi[e] || isd[e];
      if (kb + (u64)e < b && !valid[e]) firstbad = (u32)e;
    }
    /* the walk stops at the first bad op of the record: ops at or after it are dead */
    const u64 badm = __ballot(firstbad < 4u);
    u32 stop = 0xFFFFFFFFu; /* index inside this step */
    if (badm) {
      const int bl = (int)__builtin_ctzll(badm);
      stop = (u32)bl * 4u + (u32)__shfl((int)firstbad, bl);
    }
    u32 ta[4], qa[4], tsum = 0, qsum = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const bool live = (kb + (u64)e < b) && (lane * 4u + (u32)e < stop);
      ta[e] = live && !isi[e] ? len[e] : 0u;
      qa[e] = live && !isd[e] ? len[e] : 0u;
      tsum += ta[e];
      qsum += qa[e];
    }
    /* lane sums are < 2^30 and their wave prefix < 2^36: scan the two 16-bit halves (DPP) and recombine */
    u32 tl = 0, th = 0, ql = 0, qh = 0;
    u64 tp = 0, qp = 0;
    if (POS) {
      tl = wave_incl_scan_u32(tsum & 0xFFFFu), th = wave_incl_scan_u32(tsum >> 16);
      ql = wave_incl_scan_u32(qsum & 0xFFFFu), qh = wave_incl_scan_u32(qsum >> 16);
      const u64 t_incl = ((u64)th << 16) + (u64)tl, q_incl = ((u64)qh << 16) + (u64)ql;
      tp = t_base + t_incl - (u64)tsum, qp = q_base + q_incl - (u64)qsum; /* before this lane's first op */
    } else {
      acc_t += (u64)tsum;
      acc_q += (u64)qsum;
    }
    u32 prev = (u32)__shfl_up((int)code[3], 1u);
    if (lane == 0) prev = carry_code;
    u32 nxt = (u32)__shfl_down((int)code[0], 1u);
    if (lane == 63u) nxt = k0 + 256u < nops ? (rec[k0 + 256u] & 15u) : 0xFu; /* the record's next op, whoever walks it */
    bool is_ev[4];
    u32 nev = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const u32 pc = e == 0 ? prev : code[e - 1];
      const u32 nc = e == 3 ? nxt : code[e + 1];
      const bool live = (kb + (u64)e < b) && (lane * 4u + (u32)e < stop);
      const bool after_m = pc == WGA_OP_M || pc == WGA_OP_EQ || pc == WGA_OP_X;
      const bool cont_follows = nc == WGA_OP_I_CONT || nc == WGA_OP_D_CONT;
      const bool head_indel = code[e] == WGA_OP_I || code[e] == WGA_OP_D;
      is_ev[e] = live && ((code[e] == WGA_OP_X && snp) || (head_indel && after_m && ((u64)len[e] > svlen || cont_follows)));
      nev += is_ev[e] ? 1u : 0u;
    }
    if (!POS) acc_e += nev;
    const u32 einc = POS ? wave_incl_scan_u32(nev) : 0u;
    if (POS && eout && nev) {
      u64* e_out = eout + 3 * (e_base + (u64)(einc - nev));
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (is_ev[e]) {
          e_out[0] = kb + (u64)e;
          e_out[1] = tp;
          e_out[2] = qp;
          e_out += 3;
        }
        tp += ta[e];
        qp += qa[e];
      }
    }
    if (POS) {
      e_base += (u64)wave_last_u32(einc);
      t_base += ((u64)wave_last_u32(th) << 16) + (u64)wave_last_u32(tl);
      q_base += ((u64)wave_last_u32(qh) << 16) + (u64)wave_last_u32(ql);
    }
    carry_code = (u32)__shfl((int)code[3], 63);
    if (badm) {
      *bad_at = k0 + (u64)stop;
      break;
    }
  }
  if (!POS) { /* the lanes' sums, added up once */
    t_base += __shfl(wave_incl_scan_u64(acc_t, lane), 63);
    q_base += __shfl(wave_incl_scan_u64(acc_q, lane), 63);
    e_base += (u64)wave_sum_u32(acc_e);
  }
  tot->t = t_base - st.t;
  tot->q = q_base - st.q;
  tot->e = e_base - st.e;
}

/* one wave per record; records beyond `long_ops` ops are left to the piece kernels below (long_ops = 0: none are) */
__global__ __launch_bounds__(256) void k_paf_call_events(u32 n, const u32* __restrict__ ops,
                                                         const u64* __restrict__ op_off, u64 svlen,
                                                         u32 snp, u64* ev_cnt, u64* ev,
                                                         const u64* ev_off, u64 long_ops) {
  const u32 lane = threadIdx.x & 63u;
  const u64 i = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x);
  if (i >= n) return;
  const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
  if (long_ops && nops > long_ops) return;
  PafCallState z, tot;
  z.t = z.q = z.e = 0;
  u64 bad;
  if (ev) /* wave-uniform */
    paf_call_walk<true>(ops + o0, nops, 0, nops, svlen, snp, z, ev + 3 * ev_off[i], lane, &tot, &bad);
  else
    paf_call_walk<false>(ops + o0, nops, 0, nops, svlen, snp, z, (u64*)0, lane, &tot, &bad);
  if (lane == 0 && ev_cnt) ev_cnt[i] = tot.e;
}

/* ---- long records in pieces (the scheme of the MAF walks): a record beyond `long_ops` ops is cut into pieces of `piece_ops`
 *      (a multiple of 256), every piece is one wave's walk in a persistent grid over the piece list; a first walk leaves each
 *      piece's sums, one thread per record turns them into each piece's start state (running positions, events so far, "the
 *      walk has stopped": a piece behind the record's first bad op is dead), the second walk writes the events. ------------- */
__global__ __launch_bounds__(256) void k_op_piece_counts(u32 n, const u64* __restrict__ op_off, u64 long_ops, u64 piece_ops,
                                                         u32 all, u64* npieces) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 nops = op_off[i + 1] - op_off[i];
  npieces[i] = nops > long_ops ? (nops + piece_ops - 1) / piece_ops : (u64)all; /* all: the other records are one piece each */
}
struct wga_call_piece {
  u64 t, q, e;  /* MODE 0: the piece's sums; after the record scan: its start state */
  u64 bad;      /* MODE 0: record-relative first bad op or WGA_NONE; after the scan: 1 = dead, 0 = walk it */
};
/* a record's pieces are equal: its ops over its number of pieces, rounded up to whole 256-op steps (<= piece_ops) */
__device__ __forceinline__ u64 piece_span(u64 nops, u64 np) { return ((nops + np - 1) / np + 255u) & ~(u64)255u; }
/* the record of every piece (one thread per record): the walks read it instead of bisecting piece_off per piece */
__global__ __launch_bounds__(256) void k_op_piece_records(u32 n, const u64* __restrict__ piece_off, u32* piece_rec) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  for (u64 p = piece_off[i]; p < piece_off[i + 1]; p++) piece_rec[p] = i;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_paf_call_pieces(u32 n, const u32* __restrict__ ops, const u64* __restrict__ op_off,
                                                         u64 svlen, u32 snp, const u64* __restrict__ piece_off, const u32* __restrict__ piece_rec,
                                                         wga_call_piece* pc, u64* ev, const u64* ev_off) {
  const u32 lane = threadIdx.x & 63u;
  const u64 n_pieces = piece_off[n];
  const u64 n_waves = (u64)gridDim.x * 4u;
  for (u64 p = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x); p < n_pieces; p += n_waves) {
    const u32 i = WGA_UNI32(piece_rec[p]);
    const u64 o0 = op_off[i], nops = op_off[i + 1] - o0;
    const u64 psz = piece_span(nops, piece_off[i + 1] - piece_off[i]);
    const u64 a0 = (p - piece_off[i]) * psz, a = a0 < nops ? a0 : nops, b = a + psz < nops ? a + psz : nops;
    PafCallState st, tot;
    st.t = st.q = st.e = 0;
    u64 bad;
    if (MODE == 0) {
      paf_call_walk<false>(ops + o0, nops, a, b, svlen, snp, st, (u64*)0, lane, &tot, &bad);
      if (lane == 0) {
        wga_call_piece r;
        r.t = tot.t, r.q = tot.q, r.e = tot.e, r.bad = bad;
        pc[p] = r;
      }
    } else {
      const wga_call_piece r = pc[p];
      if (r.bad) continue; /* behind the record's first bad op: the reference's fold skips these ops */
      st.t = r.t, st.q = r.q, st.e = r.e;
      paf_call_walk<true>(ops + o0, nops, a, b, svlen, snp, st, ev + 3 * ev_off[i], lane, &tot, &bad);
    }
  }
}
/* one thread per long record: its pieces' sums -> start states; the record's event count */
__global__ __launch_bounds__(256) void k_paf_call_piece_scan(u32 n, const u64* __restrict__ piece_off, wga_call_piece* pc,
                                                             u64* ev_cnt) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 p0 = piece_off[i], p1 = piece_off[i + 1];
  if (p0 == p1) return;
  u64 t = 0, q = 0, e = 0;
  bool dead = false;
  for (u64 p = p0; p < p1; p++) {
    const wga_call_piece r = pc[p];
    wga_call_piece s;
    s.t = t, s.q = q, s.e = e, s.bad = dead ? 1u : 0u;
    pc[p] = s;
    if (!dead) {
      t += r.t, q += r.q, e += r.e;
      dead = r.bad != WGA_NONE;
    }
  }
  if (ev_cnt) ev_cnt[i] = e;
}

#endif /* WGA_K7_PAF_CALL_H */
