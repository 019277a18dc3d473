/*
 * wga_k3_maf.h — K3 / K4: the MAF column-pair walks (parse_maf_seq_to_cigar cigar.rs:298-308,344-432; the caller walk cigar.rs:314-328).
 * One of the parts of wga_kernels2.h, which includes them in dependency order (a part uses helpers of the parts in front of it).
 */
#ifndef WGA_K3_MAF_H
#define WGA_K3_MAF_H

#include "wga_kernels.h"

/* ============================================================================================ */
/* K3 / K4: MAF column-pair walks                                                               */
/* ============================================================================================ */
/* One wave per record, 16 columns per lane and step (one byte-unaligned 16 B load per row, 1 KiB
 * per row and wave instruction).  Columns are classified four at a time on packed bytes: a byte
 * test leaves 0x80 in every byte that satisfies it, classes are small integers kept one per byte,
 * run starts are the bytes whose class differs from the byte before (the previous lane's last
 * class comes by DPP / shuffle, the previous step's by a carried value).  Counting is popcount;
 * only the few run starts are walked bit by bit.
 *   K3  cigar_cat_ext (cigar.rs:298-308): equal bytes -> '=' (also '-','-'; case-sensitive), else
 *       target gap -> I, else query gap -> D, else X.  Run entry = start_col << 3 | class
 *       (0 '=', 1 I, 2 D, 3 X).
 *   K4  cigar_cat_ext_caller (cigar.rs:314-328): gap tests first, so '-','-' is its own class W
 *       and splits runs.  Run entry = 3 u64: start_col << 3 | class (0 '=', 1 I, 2 D, 3 X, 4 W),
 *       non-gap target characters before the run, non-gap query characters before it. */
__device__ __forceinline__ u32 zero_bytes(u32 x) { /* 0x80 in every byte of x that is 0 (exact) */
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
__device__ __forceinline__ u32 popc32(u32 x) { return (u32)__builtin_popcount(x); }

#ifndef WGA_MAF_FOLD_STEPS
#define WGA_MAF_FOLD_STEPS 4095u /* the emulator build of the tests folds every few steps instead */
#endif
#ifndef WGA_K3_BLOCKS
#define WGA_K3_BLOCKS 6 /* blocks per CU the register budget of k_maf_pair_stat is sized for (two records per wave with prefetched rows: 8 spills six registers) */
#endif
struct MafWalkOut {
  u64 ncol[5], nrun[5]; /* columns / runs per class (wave totals, valid in every lane) */
  u64 runs;             /* runs in all */
  u64 t_nongap, q_nongap; /* caller walk: non-gap characters of the two rows (including the start values) */
};

/* The walk of columns [0, L) of the rows t, q.  For a PIECE of a longer row pair the caller passes the rows advanced
 * to the piece's first column, that column's index as col_bias (reported run starts are row-relative), the class of
 * the column in front of it (carry0; 0xFF at a row start) and the non-gap characters / runs of the row in front of
 * the piece (t_base0, q_base0 for the caller walk; rout already points at the piece's first run slot). */
/* ---- sixteen columns of a lane as one bit mask: bit 8e + d = column 4d + e (dword d, byte e) ---- */
__device__ __forceinline__ u32 maf_nonzero7(u32 x) { return ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x; } /* bit 7 of every byte that is not 0 (exact) */
__device__ __forceinline__ u32 maf_gather_bit7(const u32 y[4]) {
  u32 m = (y[0] >> 7) & 0x01010101u;
  m |= (y[1] >> 6) & 0x02020202u;
  m |= (y[2] >> 5) & 0x04040404u;
  m |= (y[3] >> 4) & 0x08080808u;
  return m;
}
/* the mask moved up by one column: column j takes column j - 1's bit, column 0 takes `first` (0 / 1) */
__device__ __forceinline__ u32 maf_prev_cols(u32 b, u32 first) { return (b << 8) | ((b >> 23) & 0xEu) | first; }
/* columns [0, nv) */
__device__ __forceinline__ u32 maf_valid_mask(u32 nv) {
  u32 v = 0u;
#pragma unroll
  for (u32 d = 0; d < 4u; d++) {
    const u32 n = nv > 4u * d ? (nv - 4u * d > 4u ? 4u : nv - 4u * d) : 0u; /* valid bytes of dword d */
    const u32 low = n >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n)) - 1u);
    v |= (0x01010101u << d) & low;
  }
  return v;
}

struct MafWalkStart {
  u64 col_bias, t_base, q_base;
  u32 carry;
};

/* this lane's 16 columns of the step that starts at column c0: two byte-unaligned 16-byte loads, or byte loads in the rows'
 * last, partial vector (never beyond the rows); zeros behind the rows' end */
struct MafStepRows {
  u32 t[4], q[4];
};
__device__ __forceinline__ void maf_load_step(const u8* __restrict__ t, const u8* __restrict__ q, u64 L, u64 c0, u32 lane,
                                              MafStepRows& r) {
  const u64 c = c0 + (u64)lane * 16u;
  const u32 nv = c >= L ? 0u : (L - c >= 16u ? 16u : (u32)(L - c));
#pragma unroll
  for (int d = 0; d < 4; d++) r.t[d] = r.q[d] = 0u;
  if (nv == 16u) {
    const u32x4_a1 a = *(const u32x4_a1*)(t + c), b = *(const u32x4_a1*)(q + c);
#pragma unroll
    for (int d = 0; d < 4; d++) r.t[d] = a[d], r.q[d] = b[d];
  } else if (nv) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const u32 j = 4u * (u32)d + (u32)e;
        if (j < nv) {
          r.t[d] |= (u32)t[c + j] << (8u * (u32)e);
          r.q[d] |= (u32)q[c + j] << (8u * (u32)e);
        }
      }
    }
  }
}

/* `first`: the rows of the first step, already loaded by the caller (the kernels fetch the next record's behind the work on the
 * current one); every further step's rows are fetched one step ahead. */
template <bool CALLER>
__device__ __forceinline__ void maf_walk(const u8* __restrict__ t, const u8* __restrict__ q, u64 L,
                                         u64* rout, MafWalkOut& out, const MafWalkStart st0, const bool have_first,
                                         const MafStepRows& first) {
  const u32 lane = threadIdx.x & 63u;
  constexpr int NC = CALLER ? 5 : 4;
  /* per-lane counters of classes 1..NC-1, columns in the low and run starts in the high 16 bits (a step adds at
   * most 16 to either): folded into wave-uniform totals before they can wrap.  Class 0 needs none: its columns
   * and runs are what is left of L and of the run total. */
  u32 pk[NC];
  u64 Ctot[NC], Rtot[NC];
#pragma unroll
  for (int k = 0; k < NC; k++) pk[k] = 0u, Ctot[k] = Rtot[k] = 0ull;
  u32 carry_cls = st0.carry; /* class of the column before this step's first one */
  u64 run_base = 0, t_base = st0.t_base, q_base = st0.q_base;
  u32 steps = 0;
  u32 acc_runs = 0, acc_t = 0, acc_q = 0; /* without a run list: this lane's run starts / non-gap characters since the last fold */
  MafStepRows nx = first;
  if (!have_first) maf_load_step(t, q, L, 0, lane, nx);
  u64 c0 = 0;
  /* one step; FULL: every lane holds 16 valid columns (all steps of a row pair but the last): the validity masks and the
   * search for the last valid column fold away */
  auto step = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    const u64 c = c0 + (u64)lane * 16u;
    const u32 nv = FULL ? 16u : (c >= L ? 0u : (L - c >= 16u ? 16u : (u32)(L - c))); /* valid columns of this lane */
    u32 tw[4], qw[4];
#pragma unroll
    for (int d = 0; d < 4; d++) tw[d] = nx.t[d], qw[d] = nx.q[d];
    if (c0 + 1024 < L) maf_load_step(t, q, L, c0 + 1024, lane, nx); /* wave-uniform: the next step's rows, behind this step's work */
    /* Sixteen columns as bit masks: bit 8e + d = column 4d + e (dword d, byte e).  A class is three bit planes (b0, b1, b2:
     * I = 001, D = 010, X = 011, W = 100, '=' = 000), every test and count below is one instruction for the lane's sixteen
     * columns instead of one per dword and class. */
    u32 yn[4], yt[4], yq[4];
    const u32 any_hi = ((tw[0] | tw[1] | tw[2]) | (tw[3] | qw[0] | qw[1]) | (qw[2] | qw[3])) & 0x80808080u;
    if (__ballot(any_hi != 0u) == 0ull) { /* wave-uniform; text: no byte has bit 7, so adding 0x7F per byte cannot carry into the next */
#pragma unroll
      for (int d = 0; d < 4; d++) {
        yn[d] = (tw[d] ^ qw[d]) + 0x7F7F7F7Fu;
        yt[d] = (tw[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
        yq[d] = (qw[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
      }
    } else {
#pragma unroll
      for (int d = 0; d < 4; d++) {
        yn[d] = maf_nonzero7(tw[d] ^ qw[d]);
        yt[d] = maf_nonzero7(tw[d] ^ 0x2D2D2D2Du);
        yq[d] = maf_nonzero7(qw[d] ^ 0x2D2D2D2Du);
      }
    }
    constexpr u32 ALL = 0x0F0F0F0Fu;
    const u32 V = FULL ? ALL : maf_valid_mask(nv);
    const u32 ne = maf_gather_bit7(yn), tng = maf_gather_bit7(yt) & V, qng = maf_gather_bit7(yq) & V; /* differ; t / q hold a base */
    const u32 tg = tng ^ V, qg = qng ^ V;
    u32 b0, b1, b2 = 0u;
    if (CALLER) { /* gap tests first (cigar.rs:314-328) */
      b0 = qng & (tg | (ne & tng));
      b1 = tng & (qg | (ne & qng));
      b2 = tg & qg;
    } else { /* equal bytes first (cigar.rs:298-308): two gaps are '=' */
      b0 = ne & (tg | qng) & V;
      b1 = ne & (qg | tng) & V;
    }
    const u32 cI = b0 & ~b1, cD = b1 & ~b0, cX = b0 & b1;
    pk[1] += popc32(cI);
    pk[2] += popc32(cD);
    pk[3] += popc32(cX);
    if (CALLER) pk[NC - 1] += popc32(b2);
    /* the class of the column in front of each column: within the lane a shift of the planes, the lane's first column takes
     * the last class of the lane below (of the step / piece in front for lane 0; 0xFF / 0xFE there match no class) */
    u32 my_last;
    if (FULL) {
      my_last = ((b0 >> 27) & 1u) | ((b1 >> 26) & 2u) | ((b2 >> 25) & 4u);
    } else {
      const u32 j = nv - 1u, pos = ((j & 3u) << 3) | (j >> 2);
      my_last = nv ? (((b0 >> pos) & 1u) | (((b1 >> pos) & 1u) << 1) | (((b2 >> pos) & 1u) << 2)) : 0xFEu;
    }
    u32 prev_last = __shfl_up(my_last, 1u);
    if (lane == 0) prev_last = carry_cls;
    u32 S = (b0 ^ maf_prev_cols(b0, prev_last & 1u)) | (b1 ^ maf_prev_cols(b1, (prev_last >> 1) & 1u));
    if (CALLER)
      S |= b2 ^ maf_prev_cols(b2, (prev_last >> 2) & 1u);
    else
      S |= (prev_last >> 2) & 1u; /* a row's / piece's first column after "no class" */
    S &= V;
    const u32 nst = popc32(S);
    pk[1] += popc32(S & cI) << 16;
    pk[2] += popc32(S & cD) << 16;
    pk[3] += popc32(S & cX) << 16;
    if (CALLER) pk[NC - 1] += popc32(S & b2) << 16;
    /* ordered run list: wave-exclusive offsets of the per-lane start counts (the totals alone when nothing is written) */
    u32 step_runs, t_excl = 0, q_excl = 0, t_tot = 0, q_tot = 0;
    if (rout) {
      const u32 incl = wave_incl_scan_u32(nst);
      step_runs = wave_last_u32(incl);
      if (CALLER) {
        const u32 tnc = popc32(tng), qnc = popc32(qng);
        const u32 ti = wave_incl_scan_u32(tnc), qi = wave_incl_scan_u32(qnc);
        t_excl = ti - tnc;
        q_excl = qi - qnc;
        t_tot = wave_last_u32(ti);
        q_tot = wave_last_u32(qi);
      }
      if (nst) {
        u64 slot = run_base + (u64)(incl - nst);
        u32 tb = 0, qb = 0; /* non-gap bytes of this lane before the dword being walked */
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 m = (S >> d) & 0x01010101u; /* dword d's columns, byte e at bit 8e */
          const u32 td = (tng >> d) & 0x01010101u, qd = (qng >> d) & 0x01010101u;
          while (m) {
            const u32 bit = (u32)__builtin_ctz(m); /* 0, 8, 16 or 24 */
            const u32 pos = bit + (u32)d;
            const u32 k = ((b0 >> pos) & 1u) | (((b1 >> pos) & 1u) << 1) | (((b2 >> pos) & 1u) << 2);
            const u64 col = st0.col_bias + c + 4u * (u32)d + (bit >> 3);
            if (CALLER) {
              const u32 bm = (1u << bit) - 1u; /* bytes below */
              u64* e = rout + 3 * slot;
              e[0] = (col << 3) | (u64)k;
              e[1] = t_base + t_excl + tb + popc32(td & bm);
              e[2] = q_base + q_excl + qb + popc32(qd & bm);
            } else {
              rout[slot] = (col << 3) | (u64)k;
            }
            slot++;
            m &= m - 1u;
          }
          tb += popc32(td);
          qb += popc32(qd);
        }
      }
    } else { /* totals only: the lanes keep their own sums, added up when the walk ends (or before they could wrap) */
      step_runs = 0u;
      acc_runs += nst;
      if (CALLER) {
        acc_t += popc32(tng);
        acc_q += popc32(qng);
      }
    }
    run_base += (u64)step_runs;
    t_base += t_tot;
    q_base += q_tot;
    /* the last valid column of this step is in the last lane that has any */
    if (FULL) {
      carry_cls = wave_last_u32(my_last);
    } else {
      const u64 has = __ballot(nv != 0u);
      const int last_lane = 63 - (int)__builtin_clzll(has); /* has != 0 inside the loop */
      carry_cls = __shfl(my_last, last_lane);
    }
    if (++steps == WGA_MAF_FOLD_STEPS) { /* 16 x 4095 < 2^16: fold the lane counters before a half can wrap */
#pragma unroll
      for (int k = 1; k < NC; k++) {
        Ctot[k] += wave_sum_u32(pk[k] & 0xFFFFu);
        Rtot[k] += wave_sum_u32(pk[k] >> 16);
        pk[k] = 0u;
      }
      run_base += (u64)wave_sum_u32(acc_runs);
      t_base += (u64)wave_sum_u32(acc_t);
      q_base += (u64)wave_sum_u32(acc_q);
      acc_runs = acc_t = acc_q = 0u;
      steps = 0;
    }
  };
  for (; c0 < L; c0 += 1024) {
    if (c0 + 1024 <= L) /* wave-uniform */
      step(std::true_type{});
    else
      step(std::false_type{});
  }
  if (!rout) { /* wave-uniform */
    run_base += (u64)wave_sum_u32(acc_runs);
    if (CALLER) {
      t_base += (u64)wave_sum_u32(acc_t);
      q_base += (u64)wave_sum_u32(acc_q);
    }
  }
  /* class 0 columns / runs = all minus the others */
  u64 C[NC], R[NC];
  C[0] = L;
  R[0] = run_base;
  if (L < 65536u && L <= (u64)(WGA_MAF_FOLD_STEPS - 1u) * 1024u) { /* no fold happened and every wave total fits 16 bits: both halves in one scan */
#pragma unroll
    for (int k = 1; k < NC; k++) {
      const u32 a = wave_sum_u32(pk[k]);
      C[k] = a & 0xFFFFu, R[k] = a >> 16;
    }
  } else {
#pragma unroll
    for (int k = 1; k < NC; k++) {
      C[k] = Ctot[k] + wave_sum_u32(pk[k] & 0xFFFFu);
      R[k] = Rtot[k] + wave_sum_u32(pk[k] >> 16);
    }
  }
  u64 oc = 0, orn = 0;
#pragma unroll
  for (int k = 1; k < NC; k++) oc += C[k], orn += R[k];
  out.ncol[0] = C[0] - oc;
  out.nrun[0] = R[0] - orn;
#pragma unroll
  for (int k = 1; k < NC; k++) out.ncol[k] = C[k], out.nrun[k] = R[k];
  if (!CALLER) out.ncol[4] = out.nrun[4] = 0;
  out.runs = run_base;
  out.t_nongap = t_base;
  out.q_nongap = q_base;
}

#define WGA_MAF_PAIR_MAX 65000ull /* blocks the pair walk takes: every total fits sixteen bits */
/* ... and only where one stream is fewer steps than two walks (its lanes carry a block id, two sets of totals) */
__device__ __forceinline__ bool maf_pair_pays(u64 L0, u64 L1) {
  if (L0 >= WGA_MAF_PAIR_MAX || L1 >= WGA_MAF_PAIR_MAX) return false;
  const u64 P = (L0 + 15ull) & ~15ull;
  return (L0 + 1023ull) / 1024ull + (L1 + 1023ull) / 1024ull > (P + L1 + 1023ull) / 1024ull;
}
/* ---- two short blocks as ONE column stream -----------------------------------------------------------------------
 * A block of 1 500 columns is one full step and one of 476 columns — and a step costs its instructions whatever the number of
 * lanes that hold columns.  Two blocks of a wave are therefore walked as one stream: block A's columns, padded to a multiple of
 * sixteen (so that a lane's sixteen columns belong to ONE block), then block B's; two blocks of 1 500 columns are three steps
 * instead of four.  A lane knows its block (`rb`), its first column in it and how many of its columns are valid; the class of
 * the column in front of B's first lane is "none"; every lane keeps two sets of totals, A's and B's (a lane serves A in one step
 * and B in another); with run lists, the lanes' slots and non-gap prefixes start again at B's first lane.  The step itself — masks,
 * planes, starts, counts — is maf_walk's.  Both blocks are at most `long_cols` columns (u32 arithmetic, no folds: < 2^16 per total). */
__device__ __forceinline__ void maf_load_lane(const u8* __restrict__ tp, const u8* __restrict__ qp, u32 nv, MafStepRows& r) {
#pragma unroll
  for (int d = 0; d < 4; d++) r.t[d] = r.q[d] = 0u;
  if (nv == 16u) {
    const u32x4_a1 a = *(const u32x4_a1*)tp, b = *(const u32x4_a1*)qp;
#pragma unroll
    for (int d = 0; d < 4; d++) r.t[d] = a[d], r.q[d] = b[d];
  } else if (nv) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const u32 j = 4u * (u32)d + (u32)e;
        if (j < nv) {
          r.t[d] |= (u32)tp[j] << (8u * (u32)e);
          r.q[d] |= (u32)qp[j] << (8u * (u32)e);
        }
      }
    }
  }
}
template <bool CALLER>
__device__ __forceinline__ void maf_walk_pair(const u8* __restrict__ tA, const u8* __restrict__ qA, const u32 LA, u64* const routA,
                                              const u8* __restrict__ tB, const u8* __restrict__ qB, const u32 LB, u64* const routB,
                                              MafWalkOut& outA, MafWalkOut& outB) {
  const u32 lane = threadIdx.x & 63u;
  constexpr int NC = CALLER ? 5 : 4;
  const bool lists = routA != nullptr; /* both or neither (wave-uniform) */
  const u32 P = (LA + 15u) & ~15u, G = P + LB; /* B's first column in the stream, the stream's length */
  u32 pkA[NC], pkB[NC];                        /* per class: columns in the low, run starts in the high 16 bits */
#pragma unroll
  for (int k = 0; k < NC; k++) pkA[k] = pkB[k] = 0u;
  u32 accA = 0, accB = 0, acctA = 0, acctB = 0, accqA = 0, accqB = 0; /* without lists: run starts / non-gap characters of the lane */
  u32 runA = 0, runB = 0, tbA = 0, tbB = 0, qbA = 0, qbB = 0;         /* with lists: runs / non-gap characters in front of the step */
  u32 carry_cls = 0xFFu;
  auto lane_of = [&](u32 g0, bool& rb, u32& crel, u32& nv) { /* this lane's sixteen columns of the step at stream column g0 */
    const u32 gl = g0 + 16u * lane;
    rb = gl >= P;
    crel = rb ? gl - P : gl;
    const u32 Lr = rb ? LB : LA;
    nv = crel >= Lr ? 0u : (Lr - crel >= 16u ? 16u : Lr - crel);
  };
  MafStepRows nx;
  {
    bool rb;
    u32 crel, nv;
    lane_of(0u, rb, crel, nv);
    maf_load_lane((rb ? tB : tA) + crel, (rb ? qB : qA) + crel, nv, nx);
  }
  for (u32 g0 = 0; g0 < G; g0 += 1024u) {
    bool rb;
    u32 crel, nv;
    lane_of(g0, rb, crel, nv);
    u32 tw[4], qw[4];
#pragma unroll
    for (int d = 0; d < 4; d++) tw[d] = nx.t[d], qw[d] = nx.q[d];
    if (g0 + 1024u < G) { /* wave-uniform: the next step's rows, behind this step's work */
      bool rb2;
      u32 crel2, nv2;
      lane_of(g0 + 1024u, rb2, crel2, nv2);
      maf_load_lane((rb2 ? tB : tA) + crel2, (rb2 ? qB : qA) + crel2, nv2, nx);
    }
    u32 yn[4], yt[4], yq[4];
    const u32 any_hi = ((tw[0] | tw[1] | tw[2]) | (tw[3] | qw[0] | qw[1]) | (qw[2] | qw[3])) & 0x80808080u;
    if (__ballot(any_hi != 0u) == 0ull) {
#pragma unroll
      for (int d = 0; d < 4; d++) {
        yn[d] = (tw[d] ^ qw[d]) + 0x7F7F7F7Fu;
        yt[d] = (tw[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
        yq[d] = (qw[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
      }
    } else {
#pragma unroll
      for (int d = 0; d < 4; d++) {
        yn[d] = maf_nonzero7(tw[d] ^ qw[d]);
        yt[d] = maf_nonzero7(tw[d] ^ 0x2D2D2D2Du);
        yq[d] = maf_nonzero7(qw[d] ^ 0x2D2D2D2Du);
      }
    }
    const u32 V = maf_valid_mask(nv);
    const u32 ne = maf_gather_bit7(yn), tng = maf_gather_bit7(yt) & V, qng = maf_gather_bit7(yq) & V;
    const u32 tg = tng ^ V, qg = qng ^ V;
    u32 b0, b1, b2 = 0u;
    if (CALLER) {
      b0 = qng & (tg | (ne & tng));
      b1 = tng & (qg | (ne & qng));
      b2 = tg & qg;
    } else {
      b0 = ne & (tg | qng) & V;
      b1 = ne & (qg | tng) & V;
    }
    const u32 cI = b0 & ~b1, cD = b1 & ~b0, cX = b0 & b1;
    u32 my_last;
    {
      const u32 j = nv - 1u, pos = ((j & 3u) << 3) | (j >> 2);
      my_last = nv ? (((b0 >> pos) & 1u) | (((b1 >> pos) & 1u) << 1) | (((b2 >> pos) & 1u) << 2)) : 0xFEu;
    }
    u32 prev_last = __shfl_up(my_last, 1u);
    if (lane == 0) prev_last = carry_cls;
    if (crel == 0u) prev_last = 0xFFu; /* a block's first column */
    u32 S = (b0 ^ maf_prev_cols(b0, prev_last & 1u)) | (b1 ^ maf_prev_cols(b1, (prev_last >> 1) & 1u));
    if (CALLER)
      S |= b2 ^ maf_prev_cols(b2, (prev_last >> 2) & 1u);
    else
      S |= (prev_last >> 2) & 1u;
    S &= V;
    const u32 nst = popc32(S);
    u32 x[NC];
    x[0] = 0u;
    x[1] = popc32(cI) | (popc32(S & cI) << 16);
    x[2] = popc32(cD) | (popc32(S & cD) << 16);
    x[3] = popc32(cX) | (popc32(S & cX) << 16);
    if (CALLER) x[NC - 1] = popc32(b2) | (popc32(S & b2) << 16);
#pragma unroll
    for (int k = 1; k < NC; k++) {
      pkA[k] += rb ? 0u : x[k];
      pkB[k] += rb ? x[k] : 0u;
    }
    const u32 tnc = CALLER ? popc32(tng) : 0u, qnc = CALLER ? popc32(qng) : 0u;
    if (lists) { /* wave-uniform */
      /* the lanes of A in this step come first; B's slots and prefixes start again behind them */
      const u32 nA = g0 >= P ? 0u : ((P - g0) >> 4 > 64u ? 64u : (P - g0) >> 4);
      const u32 incl = wave_incl_scan_u32(nst);
      const u32 runs_all = wave_last_u32(incl), runs_A = nA ? wave_get_u32_dyn(incl, nA - 1u) : 0u;
      u32 t_excl = 0, q_excl = 0, t_all = 0, q_all = 0, t_A = 0, q_A = 0;
      if (CALLER) {
        const u32 ti = wave_incl_scan_u32(tnc), qi = wave_incl_scan_u32(qnc);
        t_excl = ti - tnc;
        q_excl = qi - qnc;
        t_all = wave_last_u32(ti);
        q_all = wave_last_u32(qi);
        t_A = nA ? wave_get_u32_dyn(ti, nA - 1u) : 0u;
        q_A = nA ? wave_get_u32_dyn(qi, nA - 1u) : 0u;
      }
      if (nst) {
        u64* const rout = rb ? routB : routA;
        u32 slot = (rb ? runB - runs_A : runA) + (incl - nst);
        u32 tb = (rb ? tbB - t_A : tbA) + t_excl, qb = (rb ? qbB - q_A : qbA) + q_excl;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 m = (S >> d) & 0x01010101u;
          const u32 td = (tng >> d) & 0x01010101u, qd = (qng >> d) & 0x01010101u;
          while (m) {
            const u32 bit = (u32)__builtin_ctz(m);
            const u32 pos = bit + (u32)d;
            const u32 k = ((b0 >> pos) & 1u) | (((b1 >> pos) & 1u) << 1) | (((b2 >> pos) & 1u) << 2);
            const u64 col = (u64)(crel + 4u * (u32)d + (bit >> 3));
            if (CALLER) {
              const u32 bm = (1u << bit) - 1u;
              u64* e = rout + 3 * (u64)slot;
              e[0] = (col << 3) | (u64)k;
              e[1] = (u64)(tb + popc32(td & bm));
              e[2] = (u64)(qb + popc32(qd & bm));
            } else {
              rout[slot] = (col << 3) | (u64)k;
            }
            slot++;
            m &= m - 1u;
          }
          tb += popc32(td);
          qb += popc32(qd);
        }
      }
      runA += runs_A;
      runB += runs_all - runs_A;
      tbA += t_A;
      tbB += t_all - t_A;
      qbA += q_A;
      qbB += q_all - q_A;
    } else {
      accA += rb ? 0u : nst;
      accB += rb ? nst : 0u;
      if (CALLER) {
        acctA += rb ? 0u : tnc;
        acctB += rb ? tnc : 0u;
        accqA += rb ? 0u : qnc;
        accqB += rb ? qnc : 0u;
      }
    }
    { /* the last valid column of this step is in the last lane that has any */
      const u64 has = __ballot(nv != 0u);
      if (has) carry_cls = __shfl(my_last, 63 - (int)__builtin_clzll(has));
    }
  }
  if (!lists) {
    runA = wave_sum_u32(accA);
    runB = wave_sum_u32(accB);
    if (CALLER) {
      tbA = wave_sum_u32(acctA);
      tbB = wave_sum_u32(acctB);
      qbA = wave_sum_u32(accqA);
      qbB = wave_sum_u32(accqB);
    }
  }
  u64 ocA = 0, orA = 0, ocB = 0, orB = 0;
#pragma unroll
  for (int k = 1; k < NC; k++) {
    const u32 a = wave_sum_u32(pkA[k]), b = wave_sum_u32(pkB[k]); /* < 2^16 in either half: the blocks are short */
    outA.ncol[k] = a & 0xFFFFu, outA.nrun[k] = a >> 16;
    outB.ncol[k] = b & 0xFFFFu, outB.nrun[k] = b >> 16;
    ocA += outA.ncol[k], orA += outA.nrun[k], ocB += outB.ncol[k], orB += outB.nrun[k];
  }
  outA.ncol[0] = (u64)LA - ocA, outA.nrun[0] = (u64)runA - orA;
  outB.ncol[0] = (u64)LB - ocB, outB.nrun[0] = (u64)runB - orB;
  if (!CALLER) outA.ncol[4] = outA.nrun[4] = outB.ncol[4] = outB.nrun[4] = 0;
  outA.runs = runA, outB.runs = runB;
  outA.t_nongap = tbA, outA.q_nongap = qbA, outB.t_nongap = tbB, outB.q_nongap = qbB;
}

__device__ __forceinline__ void maf_pair_store(const MafWalkOut& w, bool neg, wga_cigar_counts* cnt, u64* run_cnt, u32 lane);
__device__ __forceinline__ void maf_pair_one(const u8* __restrict__ t, const u8* __restrict__ q, u64 L, u64* rout, bool neg,
                                             const MafStepRows& first, wga_cigar_counts* cnt, u64* run_cnt, u32 lane) {
  MafWalkOut w;
  maf_walk<false>(t, q, L, rout, w, MafWalkStart{0, 0, 0, 0xFFu}, true, first);
  maf_pair_store(w, neg, cnt, run_cnt, lane);
}
__device__ __forceinline__ void maf_pair_store(const MafWalkOut& w, bool neg, wga_cigar_counts* cnt, u64* run_cnt, u32 lane) {
  /* the 11 counters leave from lanes 0..10, one field per lane (as in K1): one 88-byte store per record */
  const u64 z = 0;
  u64 v = 0;
  v = lane_put_u64<0u>(v, w.ncol[0], lane);
  v = lane_put_u64<1u>(v, w.ncol[3], lane);
  v = lane_put_u64<2u>(v, neg ? z : w.nrun[1], lane);
  v = lane_put_u64<3u>(v, neg ? z : w.ncol[1], lane);
  v = lane_put_u64<4u>(v, neg ? z : w.nrun[2], lane);
  v = lane_put_u64<5u>(v, neg ? z : w.ncol[2], lane);
  v = lane_put_u64<6u>(v, neg ? w.nrun[1] : z, lane);
  v = lane_put_u64<7u>(v, neg ? w.ncol[1] : z, lane);
  v = lane_put_u64<8u>(v, neg ? w.nrun[2] : z, lane);
  v = lane_put_u64<9u>(v, neg ? w.ncol[2] : z, lane);
  v = lane_put_u64<10u>(v, neg ? (u64)1 : z, lane);
  if (lane < 11u) ((u64*)cnt)[lane] = v;
  if (lane == 0 && run_cnt) *run_cnt = w.runs;
}

/* what a wave needs of its two blocks before their rows can be asked for: wave-uniform, scalar loads */
struct MafPairIn {
  u64 L0, L1, to0, qo0, to1, qo1, ro0, ro1;
  u32 neg0, neg1;
  bool two;
};
__device__ __forceinline__ MafPairIn maf_pair_in(u64 i0, u32 n, const u64* __restrict__ t_off, const u64* __restrict__ q_off,
                                                 const u64* __restrict__ cols, const u8* __restrict__ strand_neg,
                                                 const u64* __restrict__ run_off) {
  MafPairIn p;
  p.two = i0 + 1u < n;
  const u64 i1 = p.two ? i0 + 1u : i0;
  /* told to be wave-uniform: the values live in scalar registers across the walk */
  p.L0 = WGA_UNI64(cols[i0]), p.L1 = WGA_UNI64(cols[i1]);
  p.to0 = WGA_UNI64(t_off[i0]), p.qo0 = WGA_UNI64(q_off[i0]), p.to1 = WGA_UNI64(t_off[i1]), p.qo1 = WGA_UNI64(q_off[i1]);
  p.neg0 = strand_neg ? WGA_UNI32(strand_neg[i0] != 0 ? 1u : 0u) : 0u;
  p.neg1 = strand_neg ? WGA_UNI32(strand_neg[i1] != 0 ? 1u : 0u) : 0u;
  p.ro0 = run_off ? WGA_UNI64(run_off[i0]) : 0ull;
  p.ro1 = run_off ? WGA_UNI64(run_off[i1]) : 0ull;
  return p;
}

/* Two consecutive records per wave: the offsets of both are fetched together and the second record's first rows travel while
 * the first record is walked — three dependent round trips (offsets, rows, every further step) stood in front of the work
 * of a 1 500-column block, the step loop above and this pairing leave one.
 * (Round 5 measured a grid of resident waves that ask for the NEXT pair's offsets — scalar registers only — before they walk
 * the current pair: K3 0.344 against 0.247 ms, K4 0.249 against 0.224 on 200 000 blocks of 1 500 columns,
 * profiles/r05_maf_resident_waves_variants.txt: one wave per pair, launched by the hardware as slots free up, overlaps better.) */
__device__ __forceinline__ void maf_pair_stat_body(const MafPairIn& p, u64 i0, const u8* __restrict__ rows, wga_cigar_counts* counts,
                                                   u64* run_cnt, u64* runs, u64 long_cols, u32 lane) {
  const u64 i1 = p.two ? i0 + 1u : i0;
  const u8 *t0 = rows + p.to0, *q0 = rows + p.qo0, *t1 = rows + p.to1, *q1 = rows + p.qo1;
  const bool neg0 = p.neg0 != 0u, neg1 = p.neg1 != 0u;
  u64 *r0 = (u64*)0, *r1 = (u64*)0;
  if (runs) r0 = runs + p.ro0, r1 = runs + p.ro1;
  const u64 L0 = p.L0, L1 = p.L1;
  const bool do0 = L0 <= long_cols, do1 = p.two && L1 <= long_cols; /* a long block: walked piece by piece (k_maf_piece_walk) */
  if (do0 && do1 && maf_pair_pays(L0, L1)) { /* wave-uniform: two short blocks as one column stream */
    MafWalkOut wA, wB;
    maf_walk_pair<false>(t0, q0, (u32)L0, r0, t1, q1, (u32)L1, r1, wA, wB);
    maf_pair_store(wA, neg0, counts + i0, run_cnt ? run_cnt + i0 : (u64*)0, lane);
    maf_pair_store(wB, neg1, counts + i1, run_cnt ? run_cnt + i1 : (u64*)0, lane);
    return;
  }
  MafStepRows f0, f1;
  maf_load_step(t0, q0, do0 ? L0 : 0, 0, lane, f0);
  maf_load_step(t1, q1, do1 ? L1 : 0, 0, lane, f1);
  if (do0) maf_pair_one(t0, q0, L0, r0, neg0, f0, counts + i0, run_cnt ? run_cnt + i0 : (u64*)0, lane); /* wave-uniform */
  if (do1) maf_pair_one(t1, q1, L1, r1, neg1, f1, counts + i1, run_cnt ? run_cnt + i1 : (u64*)0, lane);
}
__global__ __launch_bounds__(256, WGA_K3_BLOCKS) void k_maf_pair_stat(u32 n, const u8* __restrict__ rows,
                                                       const u64* t_off, const u64* q_off,
                                                       const u64* cols, const u8* strand_neg,
                                                       wga_cigar_counts* counts, u64* run_cnt,
                                                       u64* runs, const u64* run_off, u64 long_cols) {
  const u32 lane = threadIdx.x & 63u;
  const u64 i0 = ((u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x)) * 2u;
  if (i0 >= n) return;
  const MafPairIn p = maf_pair_in(i0, n, t_off, q_off, cols, strand_neg, runs ? run_off : (const u64*)0);
  maf_pair_stat_body(p, i0, rows, counts, run_cnt, runs, long_cols, lane);
}

__device__ __forceinline__ void maf_call_runs_body(const MafPairIn& p, u64 i0, const u8* __restrict__ rows, u64* run_cnt, u64* runs,
                                                   u64 long_cols, u32 lane) {
  const u64 i1 = p.two ? i0 + 1u : i0;
  const u8 *t0 = rows + p.to0, *q0 = rows + p.qo0, *t1 = rows + p.to1, *q1 = rows + p.qo1;
  u64 *r0 = (u64*)0, *r1 = (u64*)0;
  if (runs) r0 = runs + 3 * p.ro0, r1 = runs + 3 * p.ro1;
  const u64 L0 = p.L0, L1 = p.L1;
  const bool do0 = L0 <= long_cols, do1 = p.two && L1 <= long_cols;
  if (do0 && do1 && maf_pair_pays(L0, L1)) { /* wave-uniform: two short blocks as one column stream */
    MafWalkOut wA, wB;
    maf_walk_pair<true>(t0, q0, (u32)L0, r0, t1, q1, (u32)L1, r1, wA, wB);
    if (lane == 0 && run_cnt) run_cnt[i0] = wA.runs, run_cnt[i1] = wB.runs;
    return;
  }
  MafStepRows f0, f1;
  maf_load_step(t0, q0, do0 ? L0 : 0, 0, lane, f0);
  maf_load_step(t1, q1, do1 ? L1 : 0, 0, lane, f1);
  if (do0) { /* wave-uniform */
    MafWalkOut w;
    maf_walk<true>(t0, q0, L0, r0, w, MafWalkStart{0, 0, 0, 0xFFu}, true, f0);
    if (lane == 0 && run_cnt) run_cnt[i0] = w.runs;
  }
  if (do1) {
    MafWalkOut w;
    maf_walk<true>(t1, q1, L1, r1, w, MafWalkStart{0, 0, 0, 0xFFu}, true, f1);
    if (lane == 0 && run_cnt) run_cnt[i1] = w.runs;
  }
}
__global__ __launch_bounds__(256) void k_maf_call_runs(u32 n, const u8* __restrict__ rows,
                                                       const u64* t_off, const u64* q_off,
                                                       const u64* cols, u64* run_cnt, u64* runs,
                                                       const u64* run_off, u64 long_cols) {
  const u32 lane = threadIdx.x & 63u;
  const u64 i0 = ((u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x)) * 2u;
  if (i0 >= n) return;
  const MafPairIn p = maf_pair_in(i0, n, t_off, q_off, cols, (const u8*)0, runs ? run_off : (const u64*)0);
  maf_call_runs_body(p, i0, rows, run_cnt, runs, long_cols, lane);
}

template <bool CALLER>
__device__ __forceinline__ void maf_walk(const u8* __restrict__ t, const u8* __restrict__ q, u64 L, u64* rout, MafWalkOut& out,
                                         const MafWalkStart st0 = MafWalkStart{0, 0, 0, 0xFFu}) {
  MafStepRows none;
#pragma unroll
  for (int d = 0; d < 4; d++) none.t[d] = none.q[d] = 0u;
  maf_walk<CALLER>(t, q, L, rout, out, st0, false, none);
}

/* ---- long blocks: the same walks, piece by piece ---------------------------------------------------------------
 * A block of 10^8 columns (`call --chunk-size` exists because such blocks do) is no work for
 * one wave.  Nothing in the walk is sequential: the class of a column is a function of that column, a run starts
 * where the class differs from the column before, the counters are sums and the caller walk's "non-gap characters
 * before the run" are prefix sums.  A block beyond `long_cols` columns is cut into pieces of `piece_cols`; every
 * piece is one wave's walk (k_maf_piece_walk, a persistent grid over the piece list), started with the class of the
 * column in front of it; a first pass leaves every piece's run and non-gap totals, an exclusive scan turns them into
 * the piece's first run slot and start values, and the fill pass writes the runs in order.  The counters of a long
 * block are added up with one atomic per field and piece. */
struct wga_maf_piece_tot {
  u64 runs, t_nongap, q_nongap;
};
/* pieces per record (0 for the records the one-wave kernels keep); long records get their counters zeroed */
__global__ __launch_bounds__(256) void k_maf_piece_counts(u32 n, const u64* cols, u64 long_cols, u64 piece_cols,
                                                          u64* npieces, wga_cigar_counts* counts, u64* run_cnt) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 c = cols[i];
  const bool lng = c > long_cols;
  npieces[i] = lng ? (c + piece_cols - 1) / piece_cols : 0;
  if (lng) {
    if (counts) {
      u64* f = (u64*)(counts + i);
      for (int k = 0; k < 11; k++) f[k] = 0;
    }
    if (run_cnt) run_cnt[i] = 0;
  }
}

__device__ __forceinline__ u32 maf_col_class(u8 tc, u8 qc, bool caller) {
  const bool tg = tc == (u8)'-', qg = qc == (u8)'-';
  if (caller) return (tg && qg) ? 4u : tg ? 1u : qg ? 2u : (tc == qc ? 0u : 3u);
  return tc == qc ? 0u : tg ? 1u : qg ? 2u : 3u;
}

struct ScanPieceTot { /* three exclusive scans in one pass over the piece totals */
  const wga_maf_piece_tot* in;
  int field;
  __device__ u64 operator()(u32 p) const { return field == 0 ? in[p].runs : field == 1 ? in[p].t_nongap : in[p].q_nongap; }
};

/* MODE 0: count (piece totals; K3 also adds the piece's counters to its record; run_cnt[i] += runs).
 * MODE 1: fill (runs written at the piece's slot).  piece_off = exclusive scan of npieces (n + 1 entries);
 * ex_runs / ex_t / ex_q = exclusive scans of the piece totals (fill only). */
template <bool CALLER, int MODE>
__global__ __launch_bounds__(256) void k_maf_piece_walk(u32 n, const u8* __restrict__ rows, const u64* t_off,
                                                        const u64* q_off, const u64* cols, const u8* strand_neg,
                                                        const u64* piece_off, u64 piece_cols,
                                                        wga_maf_piece_tot* ptot, const u64* ex_runs, const u64* ex_t,
                                                        const u64* ex_q, wga_cigar_counts* counts, u64* run_cnt,
                                                        u64* runs, const u64* run_off) {
  const u32 lane = threadIdx.x & 63u;
  const u64 n_pieces = piece_off[n];
  const u64 n_waves = (u64)gridDim.x * 4u;
  for (u64 p = (u64)blockIdx.x * 4 + WGA_WAVE_ID(threadIdx.x); p < n_pieces; p += n_waves) {
    /* record of piece p: last i with piece_off[i] <= p (wave-uniform bisection) */
    u32 lo = 0, hi = n;
    while (hi - lo > 1u) {
      const u32 mid = lo + ((hi - lo) >> 1);
      if (piece_off[mid] <= p)
        lo = mid;
      else
        hi = mid;
    }
    const u32 i = lo;
    const u64 p0 = piece_off[i];
    const u64 c0 = (p - p0) * piece_cols;
    const u64 L = cols[i] - c0 < piece_cols ? cols[i] - c0 : piece_cols;
    const u8* t = rows + t_off[i];
    const u8* q = rows + q_off[i];
    MafWalkStart st;
    st.col_bias = c0;
    st.carry = c0 ? maf_col_class(t[c0 - 1], q[c0 - 1], CALLER) : 0xFFu;
    st.t_base = st.q_base = 0;
    u64* rout = (u64*)0;
    if (MODE == 1) {
      const u64 first = ex_runs[p] - ex_runs[p0]; /* runs of this record in front of the piece */
      rout = runs + (CALLER ? 3u : 1u) * (run_off[i] + first);
      if (CALLER) {
        st.t_base = ex_t[p] - ex_t[p0];
        st.q_base = ex_q[p] - ex_q[p0];
      }
    }
    MafWalkOut w;
    maf_walk<CALLER>(t + c0, q + c0, L, rout, w, st);
    if (MODE == 0) {
      if (lane == 0) {
        wga_maf_piece_tot pt;
        pt.runs = w.runs;
        pt.t_nongap = w.t_nongap;
        pt.q_nongap = w.q_nongap;
        ptot[p] = pt;
        if (run_cnt) atomicAdd(run_cnt + i, w.runs);
      }
      if (!CALLER && counts) {
        const bool neg = strand_neg[i] != 0;
        const u64 z = 0;
        u64 v = 0;
        v = lane_put_u64<0u>(v, w.ncol[0], lane);
        v = lane_put_u64<1u>(v, w.ncol[3], lane);
        v = lane_put_u64<2u>(v, neg ? z : w.nrun[1], lane);
        v = lane_put_u64<3u>(v, neg ? z : w.ncol[1], lane);
        v = lane_put_u64<4u>(v, neg ? z : w.nrun[2], lane);
        v = lane_put_u64<5u>(v, neg ? z : w.ncol[2], lane);
        v = lane_put_u64<6u>(v, neg ? w.nrun[1] : z, lane);
        v = lane_put_u64<7u>(v, neg ? w.ncol[1] : z, lane);
        v = lane_put_u64<8u>(v, neg ? w.nrun[2] : z, lane);
        v = lane_put_u64<9u>(v, neg ? w.ncol[2] : z, lane);
        v = lane_put_u64<10u>(v, (neg && p == p0) ? (u64)1 : z, lane); /* inv_event = 1 per '-' record: its first piece */
        if (lane < 11u && v) atomicAdd((u64*)(counts + i) + lane, v);
      }
    }
  }
}

#endif /* WGA_K3_MAF_H */
