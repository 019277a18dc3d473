/*
 * wga_k3_maf.h — K3 / K4: the MAF column-pair walks (parse_maf_seq_to_cigar cigar.rs:298-308,344-432; the caller walk
 * cigar.rs:314-328), rewritten in round 6 as ONE column stream per wave.
 *
 *   K3  cigar_cat_ext (cigar.rs:298-308): equal bytes -> '=' (also '-','-'; case-sensitive), else target gap -> I, else
 *       query gap -> D, else X.  Run entry = start_col << 3 | class (0 '=', 1 I, 2 D, 3 X).
 *   K4  cigar_cat_ext_caller (cigar.rs:314-328): gap tests first, so '-','-' is its own class W and splits runs.  Run entry
 *       = 3 u64: start_col << 3 | class (0 '=', 1 I, 2 D, 3 X, 4 W), non-gap target characters before the run, non-gap
 *       query characters before it.
 *
 * What round 5's counters said about the walks they replace (profiles/r06_maf_baseline.txt): 1 100 vector + 390 scalar
 * instructions per wave of two 1 500-column blocks, 56 vector loads (48 of them single bytes of the rows' ragged ends), lanes
 * 68 % full — the kernels ran at the CU's issue ceiling, not at a memory limit, and every call paid 45 us of bookkeeping for
 * long blocks it did not have (a count kernel, a three-kernel scan and a blocking 8-byte read-back).
 *
 * The stream.  A wave takes up to eight consecutive blocks (`G`, by the batch) and walks them as one stream of columns: block
 * k's columns, padded to a multiple of 32, then block k + 1's.  A lane holds THIRTY-TWO columns of ONE block per step (two
 * 16-byte loads per row, 2 KiB per row and wave step), the step behind is in flight while a step is worked on, and a column
 * predicate of the lane is ONE 32-bit mask: bit 8e + d = column 4d + e (dword d = 0..7, byte e = 0..3 — what a shift and an
 * AND per dword produce from bit 7 of every byte).  Classes are three bit planes (I = 001, D = 010, X = 011, W = 100,
 * '=' = 000); "the class of the column in front" is one shift of a plane plus the last bit of the lane below (one DPP move);
 * run starts are where a plane differs from its shifted self; every count is a popcount.
 *   * A block's ragged end is not read byte by byte: the lane that holds it loads the block's LAST 32 bytes (they overlap the
 *     lane below) and masks the columns it shares; the overlapped bytes are the true columns in front, so the run-start test
 *     needs nothing else.  No load ever reaches outside a row.
 *   * Totals go to a per-block record in LDS with one or two LDS atomics per lane that has anything to add (a lane's
 *     sixteen-bit fields packed in u64), so a step costs the same whatever the number of blocks it touches, and the
 *     counters of the wave's blocks leave as one contiguous 88-byte-per-block store.
 *   * With run lists the lanes' slots come from one wave scan per step, taken relative to the first lane of the lane's block
 *     in this step (one bpermute) and the block's runs of earlier steps (its LDS total, read before the step adds to it).
 *   * Blocks below 32 columns (and pieces of that size) take `maf_tiny`: one column per lane, everything by ballots.
 * Long blocks (beyond `maf_long_cols`, or 61 440 columns — the sixteen-bit fields) are not walked by the stream kernel: the
 * lane that loaded such a block's length appends it to a list in the context's table, and three more launches, ALWAYS
 * queued and empty when the list is (a grid that reads one word and leaves), walk them piece by piece: a one-block plan
 * (pieces per block, their scan; the piece size grows with the total so that the table is bounded by 32 768 + n entries —
 * nothing is read back to the host, no size depends on device data), the piece walk, and for run lists a one-block scan of
 * the pieces' totals and the fill walk.  A piece is the same stream with one segment, the class of the column in front of it
 * as the carry, and the packed totals folded into 64-bit sums before a field can wrap.
 */
#ifndef WGA_K3_MAF_H
#define WGA_K3_MAF_H

#include "wga_kernels.h"

#define WGA_MAF_G 8u            /* most blocks of one wave's stream */
#define WGA_MAF_STEP 2048u      /* columns per wave step: 32 per lane */
#define WGA_MAF_SHORT_MAX 61440u /* longest block the stream kernel takes: every per-block total fits sixteen bits */
#define WGA_MAF_PIECE_BUDGET 32768u /* pieces of a call beyond one per long block: the piece size grows with the total */
#ifndef WGA_MAF_FOLD_STEPS
#define WGA_MAF_FOLD_STEPS 28u /* a piece folds its packed totals every 28 steps (57 344 columns); the emulator build of the tests every few */
#endif
#ifndef WGA_MAF_BLOCKS
#define WGA_MAF_BLOCKS 5 /* blocks per CU the register budget of the stream kernels is sized for: 96 VGPRs, no scratch.  Measured at 200 000 x 1 500 (profiles/r06_maf_blocks_per_cu.txt): 4: K3 0.116 ms, 5: 0.119, 6 (80 VGPRs, a few spilled): 0.131-0.134, 8 (64, spills in the loop): 0.220 */
#endif

__device__ __forceinline__ u32 popc32(u32 x) { return (u32)__builtin_popcount(x); }
__device__ __forceinline__ u32 maf_nonzero7(u32 x) { return ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x; } /* bit 7 of every byte that is not 0 (exact) */

/* bit 7 of every byte of eight dwords -> one mask: bit 8e + d = byte e of dword d */
__device__ __forceinline__ u32 maf_gather8(const u32 y[8]) {
  u32 m = (y[0] >> 7) & 0x01010101u;
  m |= (y[1] >> 6) & 0x02020202u;
  m |= (y[2] >> 5) & 0x04040404u;
  m |= (y[3] >> 4) & 0x08080808u;
  m |= (y[4] >> 3) & 0x10101010u;
  m |= (y[5] >> 2) & 0x20202020u;
  m |= (y[6] >> 1) & 0x40404040u;
  m |= y[7] & 0x80808080u;
  return m;
}
/* the mask moved up by one column: column j takes column j - 1's bit, column 0 takes `first` (0 / 1) */
__device__ __forceinline__ u32 maf_prev_cols(u32 b, u32 first) { return (b << 8) | ((b >> 23) & 0xFEu) | first; }
/* the columns in front of the column at bit position `bit` (= 8e + d): every column of the dwords below d, and dword d's
 * bytes below e */
__device__ __forceinline__ u32 maf_cols_before(u32 bit) {
  const u32 d = bit & 7u;
  return (((1u << d) - 1u) * 0x01010101u) | ((0x01010101u << d) & ((1u << bit) - 1u));
}
/* the columns k .. 31 (k = 0 .. 32) */
__device__ __forceinline__ u32 maf_cols_from(u32 k) {
  if (k >= 32u) return 0u;
  return ~maf_cols_before(((k & 3u) << 3) | (k >> 2));
}
__device__ __forceinline__ u32 maf_col_class(u8 tc, u8 qc, bool caller) {
  const bool tg = tc == (u8)'-', qg = qc == (u8)'-';
  if (caller) return (tg && qg) ? 4u : tg ? 1u : qg ? 2u : (tc == qc ? 0u : 3u);
  return tc == qc ? 0u : tg ? 1u : qg ? 2u : 3u;
}

/* one segment of a wave's stream (LDS): a block, or a piece of a long one */
struct MafSeg {
  u64 t, q;     /* the rows' offsets in `rows`, advanced to the segment's first column */
  u64 rout;     /* run lists: index of the segment's first run slot's first word in `runs` */
  u64 tb0, qb0; /* caller lists: non-gap characters of the block in front of the segment */
  u32 L;        /* columns walked by the stream (0: an empty, tiny or long block) */
  u32 pbeg, pend; /* the segment's place in the stream: [pbeg, pend), pend - pbeg = L rounded up to 32 */
  u32 flags;
};
#define MAF_SEG_NEG 1u   /* '-' strand (K3's counter slots) */
#define MAF_SEG_LONG 2u  /* left to the piece walk */
#define MAF_SEG_TINY 4u  /* below 32 columns: maf_tiny */
#define MAF_SEG_CONT 8u  /* a piece that continues its block: the first column has a column in front */
struct MafTot { /* what the walk leaves per segment (LDS) */
  u64 A, B;     /* K3: A = I columns | D columns << 16 | X columns << 32, B = the same for run starts (sixteen-bit fields) */
  u32 runs, tng, qng, pad;
};
struct MafRows {
  u32 t[8], q[8];
};

/* the table of a call's long blocks (context memory; see the file comment) */
struct wga_maf_long_hdr {
  u32 live_n, live_pad; /* the stream kernel's appends; the plan moves them below and clears them for the next call */
  u64 live_cols;
  u32 n_long, np;
  u64 piece_cols;
};
struct wga_maf_piece_tot {
  u64 runs, t_nongap, q_nongap;
};

/* ---- a block (or piece) below 32 columns: one column per lane, ballots -------------------------------------------------
 * carry: class of the column in front (0xFF: none).  Totals to *tot (lane 0 writes); runs to rout when given. */
template <bool CALLER>
__device__ __forceinline__ void maf_tiny(const u8* __restrict__ t, const u8* __restrict__ q, const u32 L, const u32 carry,
                                         const u64 col_bias, u64* const rout, const u64 tb0, const u64 qb0, MafTot* tot,
                                         const u32 lane) {
  const bool valid = lane < L;
  const u8 tc = valid ? t[lane] : (u8)0, qc = valid ? q[lane] : (u8)0;
  const u32 k = valid ? maf_col_class(tc, qc, CALLER) : 0xFEu;
  u32 prev = __shfl_up(k, 1u);
  if (lane == 0u) prev = carry;
  const bool start = valid && k != prev;
  const u64 S = __ballot(start);
  const u64 TN = __ballot(valid && tc != (u8)'-'), QN = __ballot(valid && qc != (u8)'-');
  const u64 mI = __ballot(valid && k == 1u), mD = __ballot(valid && k == 2u), mX = __ballot(valid && k == 3u);
  const u64 below = (1ull << lane) - 1ull;
  if (rout && start) {
    const u64 slot = (u64)__popcll(S & below);
    const u64 e0 = ((col_bias + lane) << 3) | (u64)k;
    if (CALLER) {
      u64* e = rout + 3u * slot;
      e[0] = e0;
      e[1] = tb0 + (u64)__popcll(TN & below);
      e[2] = qb0 + (u64)__popcll(QN & below);
    } else {
      rout[slot] = e0;
    }
  }
  if (lane == 0u) {
    tot->A = (u64)__popcll(mI) | ((u64)__popcll(mD) << 16) | ((u64)__popcll(mX) << 32);
    tot->B = (u64)__popcll(mI & S) | ((u64)__popcll(mD & S) << 16) | ((u64)__popcll(mX & S) << 32);
    tot->runs = (u32)__popcll(S);
    tot->tng = (u32)__popcll(TN);
    tot->qng = (u32)__popcll(QN);
  }
}

/* ---- the stream ---------------------------------------------------------------------------------------------------------
 * seg[0 .. nseg), tot[0 .. nseg) in LDS (tot zeroed, both visible to the wave); gtot = seg[nseg - 1].pend; safe_t / safe_q:
 * offsets of 32 readable bytes (a segment's first columns) for the lanes behind the stream's end, whose loads are not
 * branched around (a branch around a load makes the compiler wait for ALL loads in flight, the step being fetched
 * included: the walk would stop overlapping).
 * carry_in: planes of the column in front of the stream's first column (read when segment 0 has MAF_SEG_CONT).
 * PIECE (one segment): col_bias = the piece's first column; the K3 fields are folded into big[0 .. 6) (LDS u64: I D X
 * columns, I D X run starts) before they can wrap; the caller count pass leaves the non-gap totals in *ng_t / *ng_q. */
template <bool CALLER, bool LISTS, bool PIECE>
__device__ __forceinline__ void maf_stream(const u8* __restrict__ rows, u64* __restrict__ runs, const MafSeg* seg, MafTot* tot,
                                           const u32 nseg, const u32 gtot, const u64 safe_t, const u64 safe_q, const u32 carry_in,
                                           const u64 col_bias, u64* big, u32* ng_t, u32* ng_q) {
  const u32 lane = threadIdx.x & 63u;
  const u32 vtab = maf_cols_from(lane); /* lane k holds "columns k .. 31"; fetched by index with a bpermute */
  /* the load cursor: the segment this lane's columns of the next step to be fetched lie in */
  u32 lb = 0u;
  u64 l_t = seg[0].t, l_q = seg[0].q;
  u32 l_L = seg[0].L, l_pbeg = seg[0].pbeg, l_pend = seg[0].pend, l_cont = seg[0].flags & MAF_SEG_CONT;
  /* meta: segment | index into vtab << 8 | "the lane's first column is a block's first" << 15; cb: the block column of the
   * lane's byte 0 */
  auto fetch = [&](const u32 g0, MafRows& r, u32& meta, u32& cb) {
    const u32 g = g0 + 32u * lane;
    while (g >= l_pend && lb + 1u < nseg) { /* per lane; LDS reads only */
      lb++;
      l_t = seg[lb].t, l_q = seg[lb].q;
      l_L = seg[lb].L, l_pbeg = seg[lb].pbeg, l_pend = seg[lb].pend, l_cont = seg[lb].flags & MAF_SEG_CONT;
    }
    const bool in = g < l_pend;
    const u32 crel = g - l_pbeg, rem = l_L - crel; /* in: rem >= 1 */
    const bool tail = in && rem < 32u;             /* the block's last lane: its last 32 bytes, the shared columns masked */
    const u32 off = tail ? l_L - 32u : crel;
    const u32 vidx = !in ? 32u : tail ? 32u - rem : 0u;
    meta = lb | (vidx << 8) | ((in && crel == 0u && !l_cont) ? 0x8000u : 0u);
    cb = off;
    const u8 *tp = rows + (in ? l_t + off : safe_t), *qp = rows + (in ? l_q + off : safe_q);
    const u32x4_a1 a0 = *(const u32x4_a1*)tp, a1 = *(const u32x4_a1*)(tp + 16);
    const u32x4_a1 b0 = *(const u32x4_a1*)qp, b1 = *(const u32x4_a1*)(qp + 16);
#pragma unroll
    for (int d = 0; d < 4; d++) r.t[d] = a0[d], r.t[4 + d] = a1[d], r.q[d] = b0[d], r.q[4 + d] = b1[d];
  };
  u32 carry0 = (carry_in & 1u) << 31, carry1 = ((carry_in >> 1) & 1u) << 31, carry2 = ((carry_in >> 2) & 1u) << 31;
  u32 acc_t = 0u, acc_q = 0u, since = 0u;
  auto work = [&](const u32 g0, const MafRows& r, const u32 meta, const u32 cb) {
    const u32 blk = meta & 0xFFu;
    const u32 V = (u32)__shfl((int)vtab, (int)((meta >> 8) & 63u));
    const u32 first = (meta >> 15) & 1u;
    u32 yn[8], yt[8], yq[8];
    u32 hi = r.t[0] | r.q[0];
#pragma unroll
    for (int d = 1; d < 8; d++) hi |= r.t[d] | r.q[d];
    if (__ballot((hi & 0x80808080u) != 0u) == 0ull) { /* wave-uniform; text: no byte has bit 7, so adding 0x7F per byte cannot carry into the next */
#pragma unroll
      for (int d = 0; d < 8; d++) {
        yn[d] = (r.t[d] ^ r.q[d]) + 0x7F7F7F7Fu;
        yt[d] = (r.t[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
        yq[d] = (r.q[d] ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu;
      }
    } else {
#pragma unroll
      for (int d = 0; d < 8; d++) {
        yn[d] = maf_nonzero7(r.t[d] ^ r.q[d]);
        yt[d] = maf_nonzero7(r.t[d] ^ 0x2D2D2D2Du);
        yq[d] = maf_nonzero7(r.q[d] ^ 0x2D2D2D2Du);
      }
    }
    /* planes of all 32 bytes the lane holds (a tail lane's shared columns are real columns of the block: the column in front
     * of its first own column is among them); V masks what is counted and where runs may start */
    const u32 ne = maf_gather8(yn), tnu = maf_gather8(yt), qnu = maf_gather8(yq); /* differ; t / q hold a base */
    const u32 tng = tnu & V, qng = qnu & V;
    u32 b0, b1, b2 = 0u;
    if (CALLER) { /* gap tests first (cigar.rs:314-328) */
      b0 = qnu & (~tnu | (ne & tnu));
      b1 = tnu & (~qnu | (ne & qnu));
      b2 = ~(tnu | qnu);
    } else { /* equal bytes first (cigar.rs:298-308): two gaps are '=' */
      b0 = ne & (~tnu | qnu);
      b1 = ne & (~qnu | tnu);
    }
    /* run starts: a plane differs from the column in front (the lane below's last column for column 0; a block's first
     * column always starts a run).  A tail lane's shared columns are masked: its first own column has its true neighbour
     * in the lane itself. */
    const u32 p0 = wave_shr1_u32(b0, carry0), p1 = wave_shr1_u32(b1, carry1);
    u32 S = (b0 ^ maf_prev_cols(b0, p0 >> 31)) | (b1 ^ maf_prev_cols(b1, p1 >> 31));
    if (CALLER) {
      const u32 p2 = wave_shr1_u32(b2, carry2);
      S |= b2 ^ maf_prev_cols(b2, p2 >> 31);
    }
    S = (S | first) & V;
    carry0 = wave_get_u32(b0, 63), carry1 = wave_get_u32(b1, 63);
    if (CALLER) carry2 = wave_get_u32(b2, 63);
    const u32 nst = popc32(S);
    u32 slot0 = 0u, tb_l = 0u, qb_l = 0u;
    if (LISTS) { /* slots (and the caller walk's non-gap prefixes) relative to the block: scans over the wave, minus what the lanes in front of the block's first lane of this step hold, plus the block's earlier steps */
      const u32 pbeg = seg[blk].pbeg, pend = seg[blk].pend;
      const u32 fl = pbeg > g0 ? (pbeg - g0) >> 5 : 0u;
      const u32 incl = wave_incl_scan_u32(nst);
      const u32 fr = (u32)__shfl((int)incl, (int)(fl ? fl - 1u : 0u));
      slot0 = tot[blk].runs + (incl - nst) - (fl ? fr : 0u);
      if (CALLER) {
        const u32 tnc = popc32(tng), qnc = popc32(qng);
        const u32 ti = wave_incl_scan_u32(tnc), qi = wave_incl_scan_u32(qnc);
        const u32 ft = (u32)__shfl((int)ti, (int)(fl ? fl - 1u : 0u)), fq = (u32)__shfl((int)qi, (int)(fl ? fl - 1u : 0u));
        const u32 t_seg = ti - (fl ? ft : 0u), q_seg = qi - (fl ? fq : 0u); /* inclusive, within the block's lanes of this step */
        tb_l = tot[blk].tng + t_seg - tnc;
        qb_l = tot[blk].qng + q_seg - qnc;
        WGA_WAVE_SYNC(); /* the earlier steps' totals are read */
        const u32 g = g0 + 32u * lane;
        if (V && (lane == 63u || g + 32u >= pend)) { /* the block's last lane of this step */
          tot[blk].tng += t_seg;
          tot[blk].qng += q_seg;
        }
      } else {
        WGA_WAVE_SYNC();
      }
    } else if (CALLER && PIECE) {
      acc_t += popc32(tng);
      acc_q += popc32(qng);
    }
    if (!CALLER) {
      const u32 cI = b0 & ~b1 & V, cD = b1 & ~b0 & V, cX = b0 & b1 & V;
      if ((b0 | b1) & V) atomicAdd((unsigned long long*)&tot[blk].A, (unsigned long long)(popc32(cI) | (popc32(cD) << 16)) | ((unsigned long long)popc32(cX) << 32));
      if (nst) atomicAdd((unsigned long long*)&tot[blk].B, (unsigned long long)(popc32(S & cI) | (popc32(S & cD) << 16)) | ((unsigned long long)popc32(S & cX) << 32));
    }
    if (nst) atomicAdd(&tot[blk].runs, nst);
    if (LISTS && nst) {
      u64* const rout = runs + seg[blk].rout;
      const u64 tb0 = CALLER ? seg[blk].tb0 : 0ull, qb0 = CALLER ? seg[blk].qb0 : 0ull;
      u32 m = S;
      while (m) { /* per lane: the few run starts of its 32 columns */
        const u32 bit = (u32)__builtin_ctz(m);
        const u32 low = maf_cols_before(bit);
        const u32 k = ((b0 >> bit) & 1u) | (((b1 >> bit) & 1u) << 1) | (((b2 >> bit) & 1u) << 2);
        const u64 col = col_bias + (u64)(cb + 4u * (bit & 7u) + (bit >> 3));
        const u32 slot = slot0 + popc32(S & low);
        if (CALLER) {
          u64* e = rout + 3u * (u64)slot;
          e[0] = (col << 3) | (u64)k;
          e[1] = tb0 + (u64)(tb_l + popc32(tng & low));
          e[2] = qb0 + (u64)(qb_l + popc32(qng & low));
        } else {
          rout[slot] = (col << 3) | (u64)k;
        }
        m &= m - 1u;
      }
    }
    if (PIECE && !CALLER) { /* wave-uniform: a sixteen-bit field holds at most 65 535 */
      if (++since == WGA_MAF_FOLD_STEPS) {
        since = 0u;
        WGA_WAVE_SYNC();
        if (lane < 6u) {
          const u64 w = lane < 3u ? tot[0].A : tot[0].B;
          big[lane] += (w >> (16u * (lane % 3u))) & 0xFFFFull;
        }
        WGA_WAVE_SYNC();
        if (lane == 0u) tot[0].A = tot[0].B = 0ull;
        WGA_WAVE_SYNC();
      }
    }
  };
  /* One step is worked on while the step behind it travels: the loop takes the rows that arrived (a register copy — the one
   * place the compiler has to wait for them), asks for the next step's, and works.  Nothing in the loop branches around a
   * load: the step behind the last one is fetched like any other (its lanes are all behind the stream's end and read the safe
   * bytes), because a branch around a fetch makes the compiler wait for ALL loads in flight in front of the work (measured:
   * `s_waitcnt vmcnt(0)` right behind the fetch).  Hand-counted loads (inline asm, an s_waitcnt that pins the registers) were
   * tried and are not safe: the register allocator copies registers a load has not filled yet. */
  MafRows cur, nxt;
  u32 m_cur, c_cur, m_nxt, c_nxt;
  fetch(0u, nxt, m_nxt, c_nxt);
#pragma nounroll
  for (u32 g0 = 0u; g0 < gtot; g0 += WGA_MAF_STEP) { /* wave-uniform */
    cur = nxt, m_cur = m_nxt, c_cur = c_nxt;
    fetch(g0 + WGA_MAF_STEP, nxt, m_nxt, c_nxt);
    work(g0, cur, m_cur, c_cur);
  }
  if (CALLER && PIECE && !LISTS) {
    *ng_t = wave_sum_u32(acc_t);
    *ng_q = wave_sum_u32(acc_q);
  }
  WGA_WAVE_SYNC();
}

/* K3's eleven counters of a block from its totals (wga_cigar_counts' field order; cigar.rs:667-684 picks the plain or
 * inv_ slots by strand) */
__device__ __forceinline__ u64 maf_count_field(u32 f, u64 L, u64 cI, u64 cD, u64 cX, u64 sI, u64 sD, u64 sX, bool neg, u64 inv_event) {
  if (f == 0u) return L - cI - cD - cX;
  if (f == 1u) return cX;
  if (f == 10u) return neg ? inv_event : 0ull;
  if ((f >= 6u) != neg) return 0ull;
  const u32 g = (f - 2u) & 3u;
  return g == 0u ? sI : g == 1u ? cI : g == 2u ? sD : cD;
}

/* ---- the stream kernel: every block of the call that is not long -------------------------------------------------------- */
template <bool CALLER, bool LISTS>
__global__ __launch_bounds__(256, WGA_MAF_BLOCKS) void k_maf_stream(u32 n, u32 G, const u8* __restrict__ rows,
                                                                   const u64* __restrict__ t_off, const u64* __restrict__ q_off,
                                                                   const u64* __restrict__ cols, const u8* __restrict__ strand_neg,
                                                                   wga_cigar_counts* counts, u64* run_cnt, u64* runs,
                                                                   const u64* __restrict__ run_off, u64 long_cols,
                                                                   wga_maf_long_hdr* hdr, u32* long_list) {
  __shared__ MafSeg s_seg[4][WGA_MAF_G];
  __shared__ MafTot s_tot[4][WGA_MAF_G];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 i0 = ((u64)blockIdx.x * 4u + wave) * (u64)G;
  if (i0 >= n) return;
  const u32 nseg = WGA_UNI32((u64)n - i0 < (u64)G ? (u32)((u64)n - i0) : G);
  MafSeg* const seg = s_seg[wave];
  MafTot* const tot = s_tot[wave];
  /* lanes 0 .. nseg - 1 fetch their block's fields; a wave scan places the blocks in the stream */
  u32 Lk = 0u, fl = 0u;
  u64 to = 0, qo = 0, ro = 0, ck = 0;
  if (lane < nseg) {
    const u64 i = i0 + lane;
    ck = cols[i], to = t_off[i], qo = q_off[i];
    if (LISTS) ro = run_off[i];
    if (!CALLER && strand_neg[i] != 0) fl |= MAF_SEG_NEG;
    if (ck > long_cols || ck > (u64)WGA_MAF_SHORT_MAX) {
      fl |= MAF_SEG_LONG;
      if (hdr) { /* null: the table of the count call on these arrays is still there */
        const u32 slot = atomicAdd(&hdr->live_n, 1u);
        long_list[slot] = (u32)i;
        atomicAdd((unsigned long long*)&hdr->live_cols, (unsigned long long)ck);
      }
    } else if (ck < 32u) {
      if (ck) fl |= MAF_SEG_TINY;
    } else {
      Lk = (u32)ck;
    }
  }
  const u32 pad = (Lk + 31u) & ~31u;
  const u32 pincl = wave_incl_scan_u32(pad);
  const u32 gtot = wave_last_u32(pincl);
  if (lane < nseg) {
    MafSeg s;
    s.t = to, s.q = qo;
    s.rout = (CALLER ? 3u : 1u) * ro;
    s.tb0 = s.qb0 = 0ull;
    s.L = Lk, s.pbeg = pincl - pad, s.pend = pincl, s.flags = fl;
    seg[lane] = s;
    MafTot z;
    z.A = z.B = 0ull, z.runs = z.tng = z.qng = z.pad = 0u;
    tot[lane] = z;
  }
  WGA_WAVE_SYNC();
  if (gtot) { /* wave-uniform */
    const u32 k0 = (u32)__ffsll((unsigned long long)__ballot(Lk != 0u)) - 1u; /* a block of the stream: 32 readable bytes */
    const u64 safe_t = WGA_UNI64(seg[k0].t), safe_q = WGA_UNI64(seg[k0].q);
    maf_stream<CALLER, LISTS, false>(rows, runs, seg, tot, nseg, gtot, safe_t, safe_q, 0u, 0ull, (u64*)0, (u32*)0, (u32*)0);
  }
  if (__ballot((fl & MAF_SEG_TINY) != 0u) != 0ull) { /* wave-uniform; rare */
    for (u32 k = 0; k < nseg; k++) {
      if (!(seg[k].flags & MAF_SEG_TINY)) continue; /* wave-uniform */
      const u32 Lt = (u32)WGA_UNI64(cols[i0 + k]);
      maf_tiny<CALLER>(rows + seg[k].t, rows + seg[k].q, Lt, 0xFFu, 0ull, LISTS ? runs + seg[k].rout : (u64*)0, 0ull, 0ull, &tot[k],
                       lane);
    }
    WGA_WAVE_SYNC();
  }
  /* results: a long block's counters are cleared by the count call (the piece walk adds to them) and left alone by the fill call */
  if (!CALLER) {
    for (u32 e = lane; e < 11u * nseg; e += 64u) {
      const u32 k = e / 11u, f = e - 11u * k;
      const u32 flg = seg[k].flags;
      const bool lng = (flg & MAF_SEG_LONG) != 0u;
      const u64 A = tot[k].A, B = tot[k].B;
      const u64 Lc = lng ? 0ull : cols[i0 + k];
      const u64 v = maf_count_field(f, Lc, A & 0xFFFFull, (A >> 16) & 0xFFFFull, (A >> 32) & 0xFFFFull, B & 0xFFFFull,
                                    (B >> 16) & 0xFFFFull, (B >> 32) & 0xFFFFull, (flg & MAF_SEG_NEG) != 0u, lng ? 0ull : 1ull);
      if (!(lng && LISTS)) ((u64*)(counts + i0))[e] = v;
    }
  }
  if (run_cnt && lane < nseg && !((fl & MAF_SEG_LONG) && LISTS)) run_cnt[i0 + lane] = (u64)tot[lane].runs;
}

/* ---- long blocks ---------------------------------------------------------------------------------------------------------
 * The plan: one block of 1 024 threads; thread x owns a strip of the list.  Pieces per long block at the piece size the
 * total asks for, their exclusive scan, the counts the walks read; the live counters are cleared for the next call. */
/* exclusive scan of one value per thread over a block of 1 024 threads: a wave scan, the sixteen wave totals through LDS */
__device__ __forceinline__ u64 block_scan_1024(u64 v, u64* s /*[16]*/, u64* total) {
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const u64 inc = wave_incl_scan_u64(v, lane);
  __syncthreads(); /* s may still be read from the scan in front */
  if (lane == 63u) s[wave] = inc;
  __syncthreads();
  u64 before = 0, all = 0;
  for (u32 w = 0; w < 16u; w++) {
    const u64 x = s[w];
    if (w < wave) before += x;
    all += x;
  }
  *total = all;
  return before + inc - v;
}
__global__ __launch_bounds__(1024) void k_maf_long_plan(wga_maf_long_hdr* hdr, const u32* __restrict__ long_list,
                                                        u32* list_off, const u64* __restrict__ cols, u64 cfg_piece_cols) {
  __shared__ u64 s[16];
  const u32 x = threadIdx.x;
  const u32 nl = hdr->live_n;
  const u64 lc = hdr->live_cols;
  __syncthreads();
  if (nl == 0u) {
    if (x == 0u) hdr->n_long = 0u, hdr->np = 0u;
    return;
  }
  /* the configured piece size, unless that would make more than WGA_MAF_PIECE_BUDGET pieces (beyond one per block): then whole steps */
  u64 piece = (lc + WGA_MAF_PIECE_BUDGET - 1u) / WGA_MAF_PIECE_BUDGET;
  piece = piece <= cfg_piece_cols ? cfg_piece_cols : (piece + WGA_MAF_STEP - 1u) / WGA_MAF_STEP * WGA_MAF_STEP;
  const u32 per = (nl + 1023u) / 1024u, e0 = x * per, e1 = e0 + per < nl ? e0 + per : nl;
  u64 mine = 0;
  for (u32 e = e0; e < e1; e++) mine += (cols[long_list[e]] + piece - 1u) / piece;
  u64 total;
  u64 at = block_scan_1024(mine, s, &total);
  for (u32 e = e0; e < e1; e++) {
    list_off[e] = (u32)at;
    at += (cols[long_list[e]] + piece - 1u) / piece;
  }
  if (x == 0u) {
    list_off[nl] = (u32)total;
    hdr->n_long = nl, hdr->np = (u32)total, hdr->piece_cols = piece;
    hdr->live_n = 0u, hdr->live_cols = 0ull;
  }
}
/* exclusive scan of the pieces' totals (np + 1 entries), one block */
__global__ __launch_bounds__(1024) void k_maf_piece_scan(const wga_maf_long_hdr* hdr, const wga_maf_piece_tot* __restrict__ ptot,
                                                         wga_maf_piece_tot* ex) {
  __shared__ u64 s[16];
  const u32 x = threadIdx.x;
  const u32 np = hdr->np;
  if (np == 0u) return;
  const u32 per = (np + 1023u) / 1024u, p0 = x * per, p1 = p0 + per < np ? p0 + per : np;
  wga_maf_piece_tot mine, at;
  mine.runs = mine.t_nongap = mine.q_nongap = 0;
  u64 total;
  if (per <= 8u) { /* block-uniform: the strip in registers, its loads all in flight together (8 192 pieces: one 10^8-column block has 6 104) */
    wga_maf_piece_tot v[8];
#pragma unroll
    for (u32 k = 0; k < 8u; k++) {
      v[k].runs = v[k].t_nongap = v[k].q_nongap = 0;
      if (p0 + k < p1) v[k] = ptot[p0 + k];
    }
#pragma unroll
    for (u32 k = 0; k < 8u; k++) mine.runs += v[k].runs, mine.t_nongap += v[k].t_nongap, mine.q_nongap += v[k].q_nongap;
    at.runs = block_scan_1024(mine.runs, s, &total);
    at.t_nongap = block_scan_1024(mine.t_nongap, s, &total);
    at.q_nongap = block_scan_1024(mine.q_nongap, s, &total);
#pragma unroll
    for (u32 k = 0; k < 8u; k++) {
      if (p0 + k < p1) ex[p0 + k] = at;
      at.runs += v[k].runs, at.t_nongap += v[k].t_nongap, at.q_nongap += v[k].q_nongap;
    }
    return;
  }
  for (u32 p = p0; p < p1; p++) mine.runs += ptot[p].runs, mine.t_nongap += ptot[p].t_nongap, mine.q_nongap += ptot[p].q_nongap;
  at.runs = block_scan_1024(mine.runs, s, &total);
  at.t_nongap = block_scan_1024(mine.t_nongap, s, &total);
  at.q_nongap = block_scan_1024(mine.q_nongap, s, &total);
  for (u32 p = p0; p < p1; p++) {
    ex[p] = at;
    at.runs += ptot[p].runs, at.t_nongap += ptot[p].t_nongap, at.q_nongap += ptot[p].q_nongap;
  }
}

__device__ __forceinline__ void maf_piece_flush(wga_cigar_counts* counts, u64* run_cnt, u32 rec, u32 lane, u64 v) {
  if (lane < 11u) {
    if (counts) atomicAdd((unsigned long long*)((u64*)(counts + rec) + lane), (unsigned long long)v);
  } else if (run_cnt) {
    atomicAdd((unsigned long long*)(run_cnt + rec), (unsigned long long)v);
  }
}
/* MODE 0: count (piece totals; K3 also adds the piece's counters to its block; run_cnt[i] += runs).
 * MODE 1: fill (runs written at the piece's slot; ex = exclusive scan of the piece totals). */
template <bool CALLER, int MODE>
__global__ __launch_bounds__(256, WGA_MAF_BLOCKS) void k_maf_piece_walk(const u8* __restrict__ rows, const u64* __restrict__ t_off,
                                                                       const u64* __restrict__ q_off, const u64* __restrict__ cols,
                                                                       const u8* __restrict__ strand_neg,
                                                                       const wga_maf_long_hdr* __restrict__ hdr,
                                                                       const u32* __restrict__ long_list,
                                                                       const u32* __restrict__ list_off, wga_maf_piece_tot* ptot,
                                                                       const wga_maf_piece_tot* __restrict__ ex,
                                                                       wga_cigar_counts* counts, u64* run_cnt, u64* runs,
                                                                       const u64* __restrict__ run_off) {
  __shared__ MafSeg s_seg[4];
  __shared__ MafTot s_tot[4];
  __shared__ u64 s_big[4][6];
  __shared__ u64 s_acc[4][12]; /* MODE 0: fields 0 .. 10 of the block the wave is on, its run count */
  __shared__ u32 s_rec[4];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  u64* const acc = s_acc[wave];
  u32 acc_rec = 0xFFFFFFFFu; /* wave-uniform */
  const u32 np = hdr->np, nl = hdr->n_long;
  const u64 piece_cols = hdr->piece_cols;
  const u32 n_waves = gridDim.x * 4u;
  MafSeg* const seg = &s_seg[wave];
  MafTot* const tot = &s_tot[wave];
  u64* const big = s_big[wave];
  for (u32 p = blockIdx.x * 4u + wave; p < np; p += n_waves) { /* wave-uniform */
    /* block of piece p: last entry e with list_off[e] <= p */
    u32 lo = 0u, hi = nl;
    while (hi - lo > 1u) {
      const u32 mid = lo + ((hi - lo) >> 1);
      if (list_off[mid] <= p)
        lo = mid;
      else
        hi = mid;
    }
    const u32 pfirst = WGA_UNI32(list_off[lo]);
    const u32 i = WGA_UNI32(long_list[lo]);
    const u64 ci = WGA_UNI64(cols[i]);
    const u64 c0 = (u64)(p - pfirst) * piece_cols;
    const u32 L = (u32)(ci - c0 < piece_cols ? ci - c0 : piece_cols);
    const u64 to = WGA_UNI64(t_off[i]) + c0, qo = WGA_UNI64(q_off[i]) + c0;
    const u8* const t = rows + to;
    const u8* const q = rows + qo;
    const u32 carry = c0 ? maf_col_class(t[-1], q[-1], CALLER) : 0xFFu;
    u64 rout = 0, tb0 = 0, qb0 = 0;
    if (MODE == 1) {
      rout = (CALLER ? 3u : 1u) * (WGA_UNI64(run_off[i]) + (ex[p].runs - ex[pfirst].runs));
      if (CALLER) tb0 = ex[p].t_nongap - ex[pfirst].t_nongap, qb0 = ex[p].q_nongap - ex[pfirst].q_nongap;
    }
    WGA_WAVE_SYNC(); /* the piece in front has been read out */
    if (lane == 0u) {
      MafSeg s;
      s.t = to, s.q = qo, s.rout = rout, s.tb0 = tb0, s.qb0 = qb0;
      s.L = L, s.pbeg = 0u, s.pend = (L + 31u) & ~31u, s.flags = c0 ? MAF_SEG_CONT : 0u;
      *seg = s;
      MafTot z;
      z.A = z.B = 0ull, z.runs = z.tng = z.qng = z.pad = 0u;
      *tot = z;
    }
    if (lane < 6u) big[lane] = 0ull;
    WGA_WAVE_SYNC();
    u32 ng_t = 0u, ng_q = 0u;
    if (L < 32u) { /* wave-uniform */
      maf_tiny<CALLER>(t, q, L, carry, c0, MODE == 1 ? runs + rout : (u64*)0, tb0, qb0, tot, lane);
      WGA_WAVE_SYNC();
      ng_t = tot->tng, ng_q = tot->qng;
    } else {
      maf_stream<CALLER, MODE == 1, true>(rows, runs, seg, tot, 1u, (L + 31u) & ~31u, to, qo, carry == 0xFFu ? 0u : carry, c0, big, &ng_t,
                                          &ng_q);
    }
    if (MODE == 0) {
      const u64 nruns = (u64)tot->runs;
      if (lane == 0u) {
        wga_maf_piece_tot pt;
        pt.runs = nruns, pt.t_nongap = (u64)ng_t, pt.q_nongap = (u64)ng_q;
        ptot[p] = pt;
      }
      /* the block's counters and run count: kept per wave across its pieces (LDS), added to memory when the wave moves on to
       * another block and once per workgroup at the end — a global atomic per piece on ONE address costs 8 ns each and they
       * all come at the same time (6 104 pieces of one 10^8-column block: 50 us behind a 40 us walk) */
      if (counts || run_cnt) { /* wave-uniform */
        if (acc_rec != i) {
          if (acc_rec != 0xFFFFFFFFu && lane < 12u && acc[lane]) maf_piece_flush(counts, run_cnt, acc_rec, lane, acc[lane]);
          WGA_WAVE_SYNC();
          if (lane < 12u) acc[lane] = 0ull;
          acc_rec = i;
        }
        if (lane < 12u) {
          u64 v = nruns;
          if (lane < 11u) {
            const u64 A = tot->A, B = tot->B;
            v = CALLER ? 0ull
                       : maf_count_field(lane, (u64)L, big[0] + (A & 0xFFFFull), big[1] + ((A >> 16) & 0xFFFFull),
                                         big[2] + ((A >> 32) & 0xFFFFull), big[3] + (B & 0xFFFFull), big[4] + ((B >> 16) & 0xFFFFull),
                                         big[5] + ((B >> 32) & 0xFFFFull), strand_neg[i] != 0, p == pfirst ? 1ull : 0ull);
          }
          acc[lane] += v;
        }
      }
    }
  }
  if (MODE == 0 && (counts || run_cnt)) { /* the waves of a workgroup that ended on the same block add up first: the lowest of them speaks */
    if (lane == 0u) s_rec[wave] = acc_rec;
    __syncthreads();
    if (acc_rec != 0xFFFFFFFFu && lane < 12u) {
      bool leader = true;
      for (u32 w = 0; w < wave; w++)
        if (s_rec[w] == acc_rec) leader = false;
      if (leader) {
        u64 v = acc[lane];
        for (u32 w = wave + 1u; w < 4u; w++)
          if (s_rec[w] == acc_rec) v += s_acc[w][lane];
        if (v) maf_piece_flush(counts, run_cnt, acc_rec, lane, v);
      }
    }
  }
}

#endif /* WGA_K3_MAF_H */
