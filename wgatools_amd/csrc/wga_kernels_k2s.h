/*
 * wga_kernels_k2s.h — K2s, the STREAMING row kernel of paf2maf (`expand_variant` 3, the default): the same bytes as v1
 * (wga_kernels.h; the window kernel of rounds 3-5 was retired in round 6), i.e. parse_cigar_to_insert / cigar_unit_insert_seq
 * (cigar.rs:492-551) with reverse_complement (utils.rs:83-101) fused, written as a stream per wave instead of a block per tile.
 *
 * Why another one.  Round 4 measured what bounds the row kernels: not bytes but INSTRUCTION ISSUE.  A CU of this part retires
 * about 1.0-1.6 wave-instructions per cycle, scalar and vector together (scripts/micro/issue_rates.hip); v1 spends 322 of them
 * per kilobyte of row (2.9e9 VALU + 1.8e9 SALU per launch) and runs at 1.2 per cycle — it is at that ceiling, and so were the
 * staged, planned and window kernels of rounds 2-3 (line-complete stores, but as many instructions).  This kernel spends ~210:
 *   * a wave owns ONE row kind (target or query) of a run of consecutive tiles (a "job") and walks it as a stream: ops are
 *     taken in 256 at a time (one 16-byte load per lane, the next 256 already on their way), three wave scans give columns,
 *     gap bases and the row's gap events, which wait in a linear FIFO in LDS; no per-tile search, no block barriers;
 *   * output leaves in SUPER-STEPS of four kilobytes, each kilobyte one streaming store of the wave — whole 128-byte lines,
 *     once (v1 writes 60 % of its lines in two pieces).  The events of a super-step are counted per granule with one LDS
 *     atomic each (a byte per kilobyte in one word per lane) and ONE packed wave scan tells every lane, for its four granules,
 *     which event comes next.  A granule no gap touches (88 %) costs a table read, one byte-unaligned 16-byte buffer load and
 *     its share of the store; sixteen dashes cost nothing more;
 *   * the granules a gap touches are QUEUED, and the queue's lanes put all of a super-step's (~30) together in one round of
 *     the merge code — two windows under byte masks, further gaps in a short loop — and hand them back through LDS to the
 *     lanes that store them: the expensive path runs once per four kilobytes instead of once per kilobyte.  This happens
 *     BEFORE the plain windows are requested, so that its registers are free again (96 VGPRs: five waves per SIMD);
 *   * no calls, no barriers, every rare path inline and not unrolled.
 * An earlier form staged the source through a per-wave LDS ring filled by LDS-DMA (global_load_lds_dwordx4, counted vmcnt;
 * scripts/micro/stream_dma_copy.hip shows the mechanism at the rate of a plain copy): correct, but its 8-14 KB of LDS per wave
 * left 11-13 waves per CU, and at the issue ceiling occupancy is what hides the rest (git history).
 *
 * Coordinates.  Within a stream, C counts columns (M = X I D bases) and `cum` the row's gap bases (I for the target row, D
 * for the query row) over ALL ops since the stream started, across records (u32: a job's tiles hold < 2^24 columns each).
 * A record segment maps them affinely: column C is byte dst_seg + C of the output; a non-gap column reads the segment's
 * source buffer at S32 + (C - cum)  (forward)  or its sixteen-byte window starts at S32 - (C - cum)  (reverse complement).
 * FIFO entries are (start column, gap bases in front); a gap's length is the next entry's second word minus its own; three
 * sentinels stand behind the last one.
 *
 * What it does NOT do: tiles whose records are not "clean" (a slice longer or shorter than the CIGAR consumes: tails to
 * append, rows that stop early, String::insert_str panics), records within 32 bytes of a pool's edge and tiles wider than
 * 2^24 columns are marked by a pre-pass (k_stream_mark_*) and left to v1 (k_paf2maf_expand_list), as the window kernel
 * leaves its giant tiles there.
 */
#ifndef WGA_KERNELS_K2S_H
#define WGA_KERNELS_K2S_H

#include "wga_kernels.h"

#define WGA_S_FIFO 320u                 /* gap events waiting for their columns to be written (256 of one intake + what is left) */
#define WGA_S_MAX_TILE_COLS (1ull << 24) /* wider tiles are left to v1 */
#define WGA_S_MAX_JOB_TILES 32u          /* 32 x 2^24 columns stay below 2^31 */
#define WGA_S_SKIP 0x100u                /* wga_tile_desc::neg: the streaming kernel leaves this tile to v1 */
#define WGA_S_U 4u /* kilobytes of a row one super-step writes: the queued granules of that many share one round of the merge code */
#define WGA_S_WAVE_BYTES ((WGA_S_FIFO + 4u) * 8u + 1024u + 1024u + 256u + 272u + 16u)
#ifndef WGA_AUTO_LONG_VARIANT
#define WGA_AUTO_LONG_VARIANT 3 /* the row kernel of every other batch when "expand_variant" is -1: this one (0 = v1, for A/B builds) */
#endif

/* ---- pre-pass: which tiles the streaming kernel leaves to v1 ------------------------------------------------------- */
/* one thread per record: a record that is not clean (k_rec_desc flags 2 / 4 / 8), or whose slices lie within 32 bytes of an
 * edge of their pool (the streaming kernel's sixteen-byte windows reach over a slice's ends; v1 reads such rows byte by byte),
 * marks every tile it has ops in */
__global__ __launch_bounds__(256) void k_stream_mark_rec(u32 n, const wga_rec_desc* recs, const u64* op_off, u64 t_fa_bytes,
                                                         u64 q_fa_bytes, u8* tile_flag) {
  const u32 r = blockIdx.x * 256u + threadIdx.x;
  if (r >= n) return;
  const wga_rec_desc d = recs[r];
  const bool inside = d.t_src_off >= 32ull && d.t_src_off + d.t_src_len + 32ull <= t_fa_bytes && d.q_src_off >= 32ull &&
                      d.q_src_off + d.q_src_len + 32ull <= q_fa_bytes;
  if ((d.neg & 0xEull) == 0ull && inside) return;
  const u64 a = op_off[r], b = op_off[r + 1];
  if (b <= a) return;
  for (u64 t = a / WGA_TILE; t <= (b - 1) / WGA_TILE; t++) tile_flag[t] = 1;
}
/* one thread per tile: the flag goes into the tile's descriptor; flagged tiles are listed for v1's row emitters, tiles beyond
 * 2^31 columns for its op-serial walk.  counts[0] / list_fast: v1 fast path, counts[1] / list_slow: op-serial. */
__global__ __launch_bounds__(256) void k_stream_mark_tile(wga_tile_desc* descs, u64 nt, const u8* tile_flag, int all_slow,
                                                          u32* counts, u32* list_fast, u32* list_slow) {
  const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
  if (g >= nt) return;
  const u64 cols = descs[g].tile_cols;
  const bool slow = all_slow || cols > WGA_FAST_COL_LIMIT;
  if (slow || tile_flag[g] || cols > WGA_S_MAX_TILE_COLS) {
    descs[g].neg |= WGA_S_SKIP;
    if (slow)
      list_slow[atomicAdd(&counts[1], 1u)] = (u32)g;
    else
      list_fast[atomicAdd(&counts[0], 1u)] = (u32)g;
  }
}

/* pafpseudo (MODE 2 below): the record descriptors of its base-mode rows in the row kernel's format.  The row of record r is
 * the query slice in TARGET coordinates (M = X D columns); column x >= skip goes to out + dst_off + x - skip, so the "row
 * offset" is dst_off - skip (it may wrap below zero: only columns >= skip are written) and skip rides in the unused t_src_len.
 * A record is clean when its slice is exactly what the CIGAR consumes (M = X I S): no leftover bases to append, no
 * String::drain / insert_str panic (cigar.rs:769-786); the others, and slices within 32 bytes of the pool's edges, mark their
 * tiles for k_pafpseudo_fill_list. */
__global__ __launch_bounds__(256) void k_pseudo_rec_desc(u32 n, const wga_class_sums* sums, const u8* strand_neg, const u64* q_src_off,
                                                         const u64* q_src_len, const u64* skip, const u64* dst_off, const u64* op_off,
                                                         u64 q_fa_bytes, wga_rec_desc* out, u8* tile_flag) {
  const u32 r = blockIdx.x * 256u + threadIdx.x;
  if (r >= n) return;
  const wga_class_sums cs = sums[r];
  wga_rec_desc d;
  d.t_row_off = 0;
  d.q_row_off = dst_off[r] - skip[r];
  d.t_src_off = 0;
  d.t_src_len = skip[r];
  d.q_src_off = q_src_off ? q_src_off[r] : 0ull; /* symbol mode: no slices, every record is clean */
  d.q_src_len = q_src_len ? q_src_len[r] : 0ull;
  d.I_total = cs.i + cs.s;
  d.D_total = cs.d;
  d.L = cs.mx + cs.d;
  const bool clean = !q_src_len || d.q_src_len == cs.mx + cs.i + cs.s;
  const bool inside = !q_src_off || (d.q_src_off >= 32ull && d.q_src_off + d.q_src_len + 32ull <= q_fa_bytes);
  d.neg = (strand_neg[r] != 0 ? 1u : 0u) | (clean ? 0u : 8u);
  out[r] = d;
  if (clean && inside) return;
  const u64 a = op_off[r], b = op_off[r + 1];
  if (b <= a) return;
  for (u64 t = a / WGA_TILE; t <= (b - 1) / WGA_TILE; t++) tile_flag[t] = 1;
}

/* comp4 (wga_kernels.h) with '-' mapped to itself, so that granules that already hold gap characters pass through unchanged;
 * a '-' is still flagged as an invalid BASE (a plain granule only holds source bytes) */
__device__ __forceinline__ u32 comp4s(u32 x, u32* bad) {
  const u32 sel = x & 0x07070707u;
  const u32 fold = byte_perm(0x474EFF54u, 0x43FF41FFu, sel);
  const u32 comp = byte_perm(0x434E0D41u, 0x47FF54FFu, sel);
  *bad = (x & 0xDFDFDFDFu) ^ fold;
  return comp | (x & 0x20202020u);
}

/* ---- rare paths: small loops, never unrolled (the kernel's hot loop has to stay small: it runs at the instruction issue rate).
 *      They are inlined all the same: a call anywhere in the loop makes the compiler wait for every outstanding memory
 *      operation around it — each of a super-step's four stores then waited for the one in front. ---- */
#define WGA_S_NOINLINE __forceinline__
/* bytes [lo, hi) of a granule: byte stores, never read-modify-write */
__device__ WGA_S_NOINLINE void stream_store_bytes(u8* p, u32 o0, u32 o1, u32 o2, u32 o3, int lo, int hi) {
#pragma clang loop vectorize(disable) unroll(disable)
  for (int j = lo; j < hi; j++) {
    const u32 wj = j < 4 ? o0 : j < 8 ? o1 : j < 12 ? o2 : o3;
    p[j] = (u8)(wj >> (8 * (j & 3)));
  }
}
/* InvalidBase (utils.rs:97) in bytes [lo, hi) of the granule at column Cl: the first offender in reversed order = the smallest
 * slice index wins.  fifo == NULL: a plain granule, every byte reads the source with `adj` gap bases in front; otherwise the
 * events from k - 1 on are walked for each flagged byte. */
__device__ WGA_S_NOINLINE void stream_report_bad(u64* bad_base_pos, u32 b0, u32 b1, u32 b2, u32 b3, int lo, int hi, u32 Cl,
                                                 i64 Kseg, const u32* fifo, u32 k, u32 nf, u32 adj) {
#pragma clang loop vectorize(disable) unroll(disable)
  for (int j = lo; j < hi; j++) {
    const u32 bw = j < 4 ? b0 : j < 8 ? b1 : j < 12 ? b2 : b3;
    if (((bw >> (8 * (j & 3))) & 0xFFu) == 0u) continue;
    const u32 cj = Cl + (u32)j;
    u32 ad = adj;
    if (fifo) {
      u32 i = k - 1u; /* last event that starts at or before column cj */
      while (i + 1u < nf && (int)(fifo[2u * (i + 1u)] - cj) <= 0) i++;
      ad = fifo[2u * (i + 1u) + 1u];
    }
    atomicMin(bad_base_pos, (u64)((i64)(u64)(cj - ad) + Kseg));
  }
}

/* ---- the stream of one wave ----------------------------------------------------------------------------------------- */
/* MODE 0: paf2maf's target row (gaps = I ops), 1: its query row (gaps = D ops), 2: pafpseudo's base-mode row (K6: the query in
 * TARGET coordinates, gen_pesudo_maf_by_cigar cigar.rs:744-804 — D ops are gaps, I / S ops SKIP source bytes without writing a
 * column: an event with no gap characters that lowers the adjustment; a FIFO entry's second word is then "gap bases minus
 * skipped bases in front" and an event's gap length max(0, next - own): the sign tells the two kinds apart) */
#define WGA_S_PSEUDO 2
/* MODE 3: pafpseudo's symbol-mode row ('1' for M / =, '0' for X, '-' for D; cigar.rs:760-796): no source at all — the row is a
 * background of '1' and two kinds of "gaps", X runs and D runs.  A FIFO entry's second word is then twice the gap columns in
 * front, bit 0 = the kind of the entry's own run (1: '-'). */
#define WGA_S_SYMBOL 3
template <int MODE>
__device__ __forceinline__ u32 stream_glen(u32 c0, u32 c1) { /* gap characters of the event between two adjustments */
  return MODE == WGA_S_SYMBOL ? (c1 >> 1) - (c0 >> 1) : MODE == WGA_S_PSEUDO ? ((int)(c1 - c0) > 0 ? c1 - c0 : 0u) : c1 - c0;
}
__device__ __forceinline__ u32 stream_symbol(u32 cu) { return (cu & 1u) ? 0x2D2D2D2Du : 0x30303030u; } /* the run's character */
template <int MODE>
__device__ __forceinline__ void stream_row(const ExpandArgs& a, u8* const lds, const u32 lane, const u64 t0, const u64 t1) {
  u32* const s_fifo = (u32*)lds;                                 /* (WGA_S_FIFO + 4) x (start column, cum)    */
  u32x4_a16* const s_P = (u32x4_a16*)(s_fifo + 2u * (WGA_S_FIFO + 4u)); /* 64 granules put together by the queue's lanes */
  u32* const s_q = (u32*)(s_P + 64);                             /* the queue: (event index, granule) of up to 256 granules */
  u32* const s_T = s_q + 256;                                    /* events per granule of a super-step, a byte per kilobyte */
  u32x4_a16* const s_lm = (u32x4_a16*)(s_T + 64);                /* bytes [0, n) of a granule, n = 0 .. 16    */
  u64* const s_k = (u64*)(s_lm + 17);                            /* rarely used wave-uniform state: [0] Kseg  */
  if (lane < 17u) {
    u32x4_a16 m;
    for (int d = 0; d < 4; d++) m[d] = bytemask(0, (int)lane - 4 * d);
    s_lm[lane] = m;
  }
  constexpr bool QROW = MODE != 0;
  constexpr bool TCOORD = MODE >= WGA_S_PSEUDO; /* pafpseudo: a row in target coordinates whose head may be trimmed */
  constexpr bool SYM = MODE == WGA_S_SYMBOL;
  constexpr u32 COL_MASK = TCOORD ? 0x585u : 0x787u;                 /* op codes that are columns of the row: M D = X (+ I in paf2maf) */
  constexpr u32 GAP_MASK = MODE == 0 ? 0x202u : SYM ? 0x504u : 0x404u; /* ... that are its gaps: I / D (and their continuation codes); symbols: X D */
  constexpr u32 SKIP_MASK = MODE == WGA_S_PSEUDO ? 0x212u : 0u;      /* ... that skip source bytes: I S (pafpseudo) */
  const u8* const fa = QROW ? a.q_fa : a.t_fa;
  const u64 fa_bytes = QROW ? a.q_fa_bytes : a.t_fa_bytes;
  const u64 job_lo = t0 * WGA_TILE;
  const u32 q_end = WGA_UNI32((u32)((t1 * WGA_TILE < a.n_ops ? t1 * WGA_TILE : a.n_ops) - job_lo)); /* ops of the job */
  /* the job's ops as a buffer: lanes behind the end of the op array read zeros */
  const BufRsrc obuf = buf_make(a.ops + job_lo, (u32)(((a.n_ops - job_lo) < 0x3FFFFFFFull ? (a.n_ops - job_lo) : 0x3FFFFFFFull) * 4ull));

  /* ---- wave-uniform state (32-bit wherever the job's size allows) ---- */
  u32 C_known = 0, cum_known = 0; /* columns / gap bases of the ops taken in                      */
  u32 nf = 1;          /* FIFO entries [0, nf), sentinels at nf .. nf + 2; entry 0 lies in front  */
  u32 e0 = 1;          /* first FIFO entry whose gap starts at or behind pos                      */
  u32 pos = 0;         /* next column to write                                                    */
  u32 q = 0;           /* next op to take in (relative to the job's first op; a multiple of 256)  */
  u32 q_lo = 0, q_hi = 0; /* the ops taken in last                                                */
  u32 rec = 0, re = 0; /* the record, its last op + 1 (relative to the job, saturated)            */
  u8* dst_seg = nullptr;     /* byte of column 0 (stream coordinates) of the segment's row       */
  BufRsrc sbuf = buf_make(fa, 0u); /* the segment's source: the pool from a base a little in front of what the job can reach */
  u32 S32 = 0;         /* a non-gap column C with `cum` gap bases in front reads sbuf at S32 +/- (C - cum) (rc: its window's start) */
  bool rc = false;
  bool live = false;   /* a stream is under way                                                   */
  bool bnd = false;    /* a record ends at column C_b (gap bases cum_b in front) of the ops taken in */
  bool fin = false;    /* write everything known, even a super-step that does not fill its kilobytes */
  u32 C_b = 0, cum_b = 0;
  u32 C_min = 0;       /* pafpseudo: the record's first column that is written (its overlap with the record in front is trimmed) */
  u64 t = t0;          /* tile of op q                                                            */
  u32 ow[4] = {0, 0, 0, 0}, xl0 = 0, xg0 = 0; /* the ops taken in last, the columns / gap bases in front of each lane's first */
  u32 own[4] = {0, 0, 0, 0}; /* the next 256 ops, on their way */

  /* pafpseudo only: skip events are no columns, so any number of them can stand on ONE column (consecutive I / S ops) and
   * neither the byte counters of a super-step nor the FIFO's flush get past them.  Of a run of skip events on one column only
   * the first is needed (every entry holds the adjustment in front of it, the next entry the one behind the whole run): the
   * others are taken out, by one lane — rare and short */
  auto fifo_dedupe = [&]() {
    WGA_WAVE_SYNC();
    u32 n_new = nf;
    if (lane == 0u) {
      u32 w = e0;
      for (u32 r = e0; r < nf; r++) {
        const u32 gs = s_fifo[2u * r], cu = s_fifo[2u * r + 1u];
        const bool dup = w > e0 && s_fifo[2u * (w - 1u)] == gs && (int)(cu - s_fifo[2u * (w - 1u) + 1u]) <= 0 &&
                         (int)(s_fifo[2u * (r + 1u) + 1u] - cu) <= 0; /* the one kept in front is a skip on this column, and so is this */
        if (!dup) {
          s_fifo[2u * w] = gs;
          s_fifo[2u * w + 1u] = cu;
          w++;
        }
      }
      for (u32 k = 0; k < 3u; k++) { /* the sentinels move up */
        s_fifo[2u * (w + k)] = s_fifo[2u * (nf + k)];
        s_fifo[2u * (w + k) + 1u] = s_fifo[2u * (nf + k) + 1u];
      }
      n_new = w;
    }
    nf = wave_get_u32(n_new, 0);
    WGA_WAVE_SYNC();
  };

  for (;;) {
    /* ================= write columns: ONE site for the super-step ================= */
    {
      const u32 lim = bnd ? C_b : C_known;
      const bool all = bnd || fin;
      while ((int)(lim - pos) > 0) {
        if (TCOORD && (int)(C_min - pos) > 0) { /* the record's columns in front of its first written one */
          const u32 to = (int)(lim - C_min) > 0 ? C_min : lim;
          u32 c;
          do { /* e0: past the events that start in front of `to` */
            const u32 idx = e0 + lane < nf ? e0 + lane : nf;
            c = (u32)__popcll(__ballot(idx < nf && (int)(s_fifo[2u * idx] - to) < 0));
            e0 += c;
          } while (c == 64u);
          pos = to;
          continue;
        }
        u8* const A = dst_seg + pos;
        const u32 mis = (u32)(u64)A & 1023u;
        u8* const B = A - mis;
        const u32 room = WGA_S_U * 1024u - mis;
        if (!all && (lim - pos) < room) break;
        const u32 Cs = pos;
        u32 Ce = (lim - pos) < room ? lim : pos + room;
        const u32 Cl0 = Cs - mis; /* column of the first granule of the kilobyte Cs lies in (wraps below zero at a stream's start) */
        if (e0 + 255u < nf) { /* gaps every few columns: a super-step counts its events in bytes, so it ends in front of its 256th */
          const u32 g255 = WGA_UNI32(s_fifo[2u * (e0 + 255u)]);
          if ((int)(g255 - Ce) < 0) Ce = Cl0 + ((g255 - Cl0) & ~15u);
          if (MODE == WGA_S_PSEUDO && (int)(Ce - Cs) < 16) { /* hundreds of skip events on one column: made one, then again */
            fifo_dedupe();
            continue;
          }
        }

        /* ---- (1) the super-step's events, counted per granule: byte u of T[l] = events that start in granule l of kilobyte u ---- */
        s_T[lane] = 0u;
        WGA_WAVE_SYNC();
        u32 nE = 0;
        {
          u32 base = e0, c;
          do {
            const u32 idx = base + lane < nf ? base + lane : nf;
            const u32 gsr = s_fifo[2u * idx] - Cl0;
            const bool in = gsr < Ce - Cl0;
            c = (u32)__popcll(__ballot(in));
            if (in) atomicAdd(&s_T[(gsr >> 4) & 63u], 1u << (8u * (gsr >> 10)));
            base += c;
            nE += c;
          } while (c == 64u);
        }
        WGA_WAVE_SYNC();
        const u32 tw = s_T[lane];
        WGA_WAVE_SYNC();
        /* one scan for all kilobytes (no byte overflows: at most 255 events); events in front of a granule, in granule order */
        const u32 tinc = wave_incl_scan_u32(tw);
        const u32 texc = tinc - tw;
        const u32 ttot = wave_last_u32(tinc);

        /* ---- (2) every lane's granule of every kilobyte is classified; the ones a gap touches are queued ---- */
        u32 kbase = e0; /* the first event of the kilobyte */
        u32 qn = 0, rpk = 0, cls = 0; /* cls: per kilobyte u bits 4u .. 4u + 2: plain, dashes, queued */
        u32 adjv[WGA_S_U];
        const u32 wsp = Ce - Cs - 16u, asp = Ce - Cs + 15u; /* whole: (Cl - Cs) <= wsp; active: (Cl + 15 - Cs) < asp (unsigned) */
        const bool any16 = (Ce - Cs) >= 16u;
        u32 qcum = 0; /* queued granules of the kilobytes in front, a byte each (for kilobyte u: bits 8u ..) */
#pragma unroll
        for (u32 u = 0; u < WGA_S_U; u++) { /* kilobytes behind the super-step's end: no lane is active, nothing is queued */
          const u32 cnt = (tw >> (8u * u)) & 255u;
          const u32 k = kbase + ((texc >> (8u * u)) & 255u); /* the first event that starts in this granule or behind it */
          kbase += (ttot >> (8u * u)) & 255u;
          const u32 Cl = Cl0 + 1024u * u + 16u * lane;
          const bool whole = any16 && (Cl - Cs) <= wsp, active = (Cl + 15u - Cs) < asp;
          const u32x4_a1 fe = *(const u32x4_a1*)(s_fifo + 2u * (k - 1u)); /* events k - 1, k: start, gap bases in front */
          const int rel0 = (int)(fe[0] + stream_glen<MODE>(fe[1], fe[3]) - Cl); /* how far the gap in front reaches into the granule */
          const bool quiet = whole && cnt == 0u;    /* a whole granule in which no gap starts ... */
          const bool plain = quiet && rel0 <= 0;    /* ... sixteen source bytes in a row        */
          const bool dashes = quiet && rel0 >= 16;  /* ... sixteen gap characters               */
          const bool flagged = active && !(plain || dashes);
          adjv[u] = fe[3];
          cls |= ((plain ? 1u : 0u) | (dashes ? 2u : 0u) | (flagged ? 4u : 0u) | (SYM ? (fe[1] & 1u) << 3 : 0u)) << (4u * u);
          const u64 m = __ballot(flagged);
          if (flagged) {
            const u32 r = qn + lane_rank(m, lane);
            s_q[r & 255u] = (k << 8) | (u << 6) | lane;
            rpk |= (r & 255u) << (8u * u);
          }
          qcum |= (qn > 255u ? 255u : qn) << (8u * u);
          qn += (u32)__popcll(m);
        }
        /* the queue's lanes put the queued granules together in ONE round: a super-step ends in front of the kilobyte with
         * which it would queue more than 64 (a kilobyte has 64 granules, so at least one always stays) */
        if (qn > 64u) {
          u32 keep = 1u;
#pragma unroll
          for (u32 u = 2; u <= WGA_S_U; u++) { /* kilobytes [0, u) queue (qcum byte u, or qn for all of them) granules */
            const u32 upto = u == WGA_S_U ? qn : (qcum >> (8u * u)) & 255u;
            const bool sat = u < WGA_S_U && ((qcum >> (8u * u)) & 255u) == 255u; /* the byte saturated: more than 64 anyway */
            if (upto <= 64u && !sat) keep = u;
          }
          const u32 Cn = Cl0 + 1024u * keep;
          if ((int)(Cn - Ce) < 0) Ce = Cn;
          qn = keep == WGA_S_U ? qn : (qcum >> (8u * keep)) & 255u;
          nE = 0;
#pragma unroll
          for (u32 u = 0; u < WGA_S_U; u++)
            if (u < keep) nE += (ttot >> (8u * u)) & 255u;
        }
        const BufRsrc dbuf = buf_make(B, WGA_S_U * 1024u);

        /* ---- (3) the queued granules: put together from two windows under byte masks, handed back through LDS ---- */
        WGA_WAVE_SYNC();
        if (lane < qn) {
          const u32 ent = s_q[lane];
          const u32 k = ent >> 8, Cl = Cl0 + 16u * (ent & 255u);
          int lo = (int)(Cs - Cl), hi = (int)(Ce - Cl);
          lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
          hi = hi > 16 ? 16 : (hi < 0 ? 0 : hi);
          const u32x4_a1 ea = *(const u32x4_a1*)(s_fifo + 2u * (k - 1u)); /* events k - 1, k */
          const u32x4_a1 eb = *(const u32x4_a1*)(s_fifo + 2u * (k + 1u)); /* events k + 1, k + 2 */
          const u32 gs0 = ea[0], cu0 = ea[1], gs1 = ea[2], cu1 = ea[3], gs2 = eb[0], cu2 = eb[1];
          /* [lo, a1) the rest of the gap in front | [a1, b1) source behind it | [b1, e1) the gap that starts here | [e1, hi) source */
          int a1 = (int)(gs0 + stream_glen<MODE>(cu0, cu1) - Cl);
          a1 = a1 < lo ? lo : (a1 > hi ? hi : a1);
          int b1 = (int)(gs1 - Cl);
          b1 = b1 < a1 ? a1 : (b1 > hi ? hi : b1);
          const u32 gl1 = stream_glen<MODE>(cu1, cu2);
          int e1 = gl1 >= 16u ? 16 : b1 + (int)gl1;
          e1 = e1 > hi ? hi : e1;
          bool more = (int)(gs2 - Cl) < hi;
          u32 W0[4], W1[4];
          if (!SYM) {
            buf_load16(sbuf, b1 > a1 ? (rc ? S32 - Cl + cu1 : S32 + Cl - cu1) : WGA_BUF_OOB, W0);
            buf_load16(sbuf, hi > e1 ? (rc ? S32 - Cl + cu2 : S32 + Cl - cu2) : WGA_BUF_OOB, W1);
          }
          const u32x4_a16 La = s_lm[a1], Lb = s_lm[b1], Le = s_lm[e1];
          u32 o[4], bad[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int d = 0; d < 4; d++) {
            if (SYM) { /* the run in front | '1' | the run that starts here | '1' */
              o[d] = bfi_b32(La[d], stream_symbol(cu0), bfi_b32(Le[d] & ~Lb[d], stream_symbol(cu1), 0x31313131u));
              continue;
            }
            u32 cp;
            if (rc)
              cp = comp4s(bfi_b32(Lb[d], bswap32(W0[3 - d]), bswap32(W1[3 - d])), &bad[d]);
            else
              cp = bfi_b32(Lb[d], W0[d], W1[d]);
            const u32 md = La[d] | (Le[d] & ~Lb[d]);
            o[d] = bfi_b32(md, 0x2D2D2D2Du, cp);
            bad[d] &= ~md;
          }
          u32 ie = k + 1u;
          while (more) { /* further events inside the granule: one round per event (rare: gaps are ~150 columns apart) */
            const u32 gsi = s_fifo[2u * ie], cui = s_fifo[2u * ie + 1u], gsn = s_fifo[2u * ie + 2u], cun = s_fifo[2u * ie + 3u];
            const int b = (int)(gsi - Cl);
            const u32 len = stream_glen<MODE>(cui, cun);
            const int e = len >= (u32)(hi - b) ? hi : b + (int)len;
            u32 W[4];
            if (SYM)
              W[0] = W[1] = W[2] = W[3] = 0x31313131u;
            else
              buf_load16(sbuf, rc ? S32 - Cl + cun : S32 + Cl - cun, W);
            const u32x4_a16 Mb = s_lm[b], Me = s_lm[e];
#pragma unroll
            for (int d = 0; d < 4; d++) {
              u32 x = W[d], bw = 0u;
              if (rc) x = comp4s(bswap32(W[3 - d]), &bw);
              if (SYM)
                o[d] = bfi_b32(Mb[d], o[d], bfi_b32(Me[d], stream_symbol(cui), x));
              else
                o[d] = bfi_b32(Mb[d], o[d], bfi_b32(Me[d], 0x2D2D2D2Du, x));
              bad[d] = (bad[d] & Mb[d]) | (bw & ~Me[d]);
            }
            ie++;
            more = (int)(gsn - Cl) < hi;
          }
          if (rc && (bad[0] | bad[1] | bad[2] | bad[3]) != 0u) /* InvalidBase */
            stream_report_bad((u64*)&a.diag[rec].bad_base_pos, bad[0], bad[1], bad[2], bad[3], lo, hi, Cl, (i64)s_k[0], s_fifo, k, nf, 0u);
          const u32x4_a16 ov = {o[0], o[1], o[2], o[3]};
          s_P[lane] = ov;
        }
        WGA_WAVE_SYNC();

        /* ---- (4) the plain granules' windows, all requested at once; what the queue made replaces the queued ones; whole
         *      granules leave in one streaming store of the wave per kilobyte, partial ones (at most two) in byte stores ---- */
        u32 dat[WGA_S_U][4];
        const u32 wsp2 = Ce - Cs - 16u; /* Ce may have moved */
#pragma unroll
        for (u32 u = 0; u < WGA_S_U; u++) {
          const u32 Cl = Cl0 + 1024u * u + 16u * lane;
          const u32 c = (cls >> (4u * u)) & 15u;
          if (SYM) { /* sixteen of the run's character, or of '1' */
            dat[u][0] = dat[u][1] = dat[u][2] = dat[u][3] = (c & 2u) ? ((c & 8u) ? 0x2D2D2D2Du : 0x30303030u) : 0x31313131u;
            continue;
          }
          dat[u][0] = dat[u][1] = dat[u][2] = dat[u][3] = (c & 2u) ? 0x2D2D2D2Du : 0u;
          if ((c & 1u) && (int)(Cl - Ce) < 0) buf_load16(sbuf, rc ? S32 - Cl + adjv[u] : S32 + Cl - adjv[u], dat[u]);
        }
        if (rc) { /* the windows arrive: reverse complement ('-' and what the queue replaces pass through) */
#pragma unroll
          for (u32 u = 0; u < WGA_S_U; u++) {
            u32 bad[4], x[4];
#pragma unroll
            for (int d = 0; d < 4; d++) x[d] = comp4s(bswap32(dat[u][3 - d]), &bad[d]);
#pragma unroll
            for (int d = 0; d < 4; d++) dat[u][d] = x[d];
            /* InvalidBase in a plain granule (utils.rs:97); queued granules are checked where they are put together */
            const u32 Cl = Cl0 + 1024u * u + 16u * lane;
            const bool pb = ((cls >> (4u * u)) & 1u) != 0u && (int)(Cl - Ce) < 0 && (bad[0] | bad[1] | bad[2] | bad[3]) != 0u;
            if (__ballot(pb)) {
              if (pb)
                stream_report_bad((u64*)&a.diag[rec].bad_base_pos, bad[0], bad[1], bad[2], bad[3], 0, 16, Cl, (i64)s_k[0], nullptr, 0u, 0u,
                                  adjv[u]);
            }
          }
        }
#pragma unroll
        for (u32 u = 0; u < WGA_S_U; u++) {
          const u32 Cl = Cl0 + 1024u * u + 16u * lane;
          const bool queued = ((cls >> (4u * u)) & 4u) != 0u && (int)(Cl - Ce) < 0;
          if (queued) {
            const u32x4_a16 pv = s_P[(rpk >> (8u * u)) & 255u];
            dat[u][0] = pv[0];
            dat[u][1] = pv[1];
            dat[u][2] = pv[2];
            dat[u][3] = pv[3];
          }
          const bool whole = (Ce - Cs) >= 16u && (Cl - Cs) <= wsp2;
          buf_store16_stream(dbuf, whole ? 1024u * u + 16u * lane : WGA_BUF_OOB, dat[u]);
          if (queued && !whole) {
            int lo = (int)(Cs - Cl), hi = (int)(Ce - Cl);
            lo = lo < 0 ? 0 : lo;
            hi = hi > 16 ? 16 : hi;
            stream_store_bytes(B + 1024u * u + lane * 16u, dat[u][0], dat[u][1], dat[u][2], dat[u][3], lo, hi);
          }
        }
        WGA_WAVE_SYNC(); /* the patch buffer and the queue are rewritten by the next super-step */
        e0 += nE;
        pos = Ce;
      }
      fin = false;
    }

    /* ================= a record ended: the next one starts at column C_b ================= */
    if (bnd || !live) {
      const bool sw = bnd; /* a switch to the next record (otherwise: a stream starts) */
      u64 row_off, src_off, src_len, x_a, sb;
      bool neg;
      u32 C_a, cum_a;
      if (bnd) {
        const u64 bnd_op = job_lo + re;
        u64 ren;
        do { /* records without ops have no rows here (v1 walks past them as well) */
          rec++;
          ren = WGA_UNI64(a.op_off[rec + 1]);
        } while (ren <= bnd_op);
        re = ren - job_lo > (u64)q_end ? q_end : (u32)(ren - job_lo);
        const wga_rec_desc* const rd = a.recs + rec;
        row_off = WGA_UNI64(QROW ? rd->q_row_off : rd->t_row_off);
        src_off = WGA_UNI64(QROW ? rd->q_src_off : rd->t_src_off);
        src_len = WGA_UNI64(QROW ? rd->q_src_len : rd->t_src_len);
        neg = (WGA_UNI64(rd->neg) & 1ull) != 0ull;
        x_a = 0;
        sb = 0;
        C_a = C_b;
        cum_a = cum_b;
        if (TCOORD) { /* the columns in front of `skip` (kept in the descriptor's unused t_src_len) are not written */
          const u64 skip = WGA_UNI64(rd->t_src_len);
          C_min = C_a + (u32)(skip < 0x40000000ull ? skip : 0x40000000ull);
        }
      } else {
        /* (re)start: the next tile that is this kernel's */
        while (t < t1 && (WGA_UNI32(a.tdesc[t].neg) & WGA_S_SKIP)) t++;
        if (t >= t1) break;
        const wga_tile_desc* const td = a.tdesc + t;
        const u64 tile_start = t * WGA_TILE;
        q = (u32)(tile_start - job_lo);
        WGA_WAVE_SYNC();
        C_known = cum_known = 0u;
        pos = 0u;
        nf = 1u;
        e0 = 1u;
        if (lane < 4u) {
          s_fifo[2u * lane] = 0u;
          s_fifo[2u * lane + 1u] = 0u;
        }
        rec = WGA_UNI32(td->rec);
        const u64 ren = WGA_UNI64(td->re), rs = WGA_UNI64(td->rs);
        re = ren - job_lo > (u64)q_end ? q_end : (u32)(ren - job_lo);
        const bool cont = rs < tile_start; /* the record began in an earlier tile */
        const u64 b_mx = cont ? WGA_UNI64(td->b_mx) : 0ull, b_i = cont ? WGA_UNI64(td->b_i) : 0ull,
                  b_d = cont ? WGA_UNI64(td->b_d) : 0ull;
        row_off = WGA_UNI64(QROW ? td->q_row_off : td->t_row_off);
        src_off = WGA_UNI64(QROW ? td->q_src_off : td->t_src_off);
        src_len = WGA_UNI64(QROW ? td->q_src_len : td->t_src_len);
        neg = (WGA_UNI32(td->neg) & 1u) != 0u;
        x_a = TCOORD ? b_mx + b_d : b_mx + b_i + b_d; /* pafpseudo: I (and S, folded into b_i) are no columns */
        sb = QROW ? b_mx + b_i : b_mx + b_d;
        C_a = cum_a = 0u;
        if (TCOORD) {
          const u64 skip = WGA_UNI64(td->t_src_len);
          const u64 ahead = skip > x_a ? skip - x_a : 0ull;
          C_min = (u32)(ahead < 0x40000000ull ? ahead : 0x40000000ull);
        }
        buf_load16(obuf, q * 4u + lane * 16u, own); /* the first 256 ops */
        live = true;
      }
      /* the segment: column C_a (cum_a gap bases in front) is column x_a of the record's row, sb bases of its slice used.
       * A non-gap column C (cum gap bases in front) reads slice index (C - cum) + Kseg, Kseg = sb - (C_a - cum_a); the job
       * reaches at most 2^30 bases further, so the buffer starts a little in front of the first one (offsets stay 32-bit). */
      dst_seg = (u8*)((u64)a.out + row_off + x_a - (u64)C_a); /* pafpseudo: row_off may have wrapped below zero */
      rc = QROW && !SYM && neg;
      const u32 adv = C_a - cum_a;
      u64 base; /* pool offset of the buffer's first byte */
      if (rc) { /* slice index s is pool byte src_off + src_len - 1 - s; a window of sixteen starts fifteen bytes below */
        const u64 top = src_off + src_len; /* one behind the pool byte of slice index 0 */
        const u64 hiB = top > sb ? top - sb : 0ull; /* one behind the first byte this segment reads */
        base = hiB > 0x60000000ull ? hiB - 0x60000000ull : 0ull;
        S32 = (u32)(top - base) - 16u - (u32)sb + adv; /* window start of column C: S32 - (C - cum) */
      } else {
        const u64 loB = src_off + sb;
        base = loB > 64ull ? loB - 64ull : 0ull;
        S32 = (u32)(loB - base) - adv; /* byte of column C: S32 + (C - cum) */
      }
      base = base > fa_bytes ? fa_bytes : base;
      {
        const u64 left = fa_bytes - base;
        sbuf = buf_make(fa + base, left > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (u32)left);
      }
      if (lane == 0u) s_k[0] = (u64)((i64)sb - (i64)(u64)adv);
      WGA_WAVE_SYNC();
      /* the next boundary among the ops that are in (none right after a start) */
      bnd = false;
      if (sw && re < q_hi) {
        const u32 kb = re - q_lo, e = kb & 3u; /* the columns / gap bases in front of op kb: its lane's prefix + its ops in front */
        u32 vl = xl0, vg = xg0;
#pragma unroll
        for (u32 i = 0; i < 3u; i++) {
          const bool valid = q_lo + 4u * lane + i < q_hi && i < e;
          const u32 len = valid ? ow[i] >> 4 : 0u, code = ow[i] & 15u;
          vl += len & bit_mask(COL_MASK, code);
          vg += (len & bit_mask(GAP_MASK, code)) - (len & bit_mask(SKIP_MASK, code));
        }
        C_b = wave_get_u32_dyn(vl, kb >> 2);
        cum_b = wave_get_u32_dyn(vg, kb >> 2);
        bnd = true;
      }
      continue;
    }

    /* ================= everything that is in has been written as far as it may: take more ops in ================= */
    if (q >= q_end) { /* the job ends: the last, partial kilobytes */
      if (pos != C_known) {
        fin = true;
        continue;
      }
      break;
    }
    if ((q & (WGA_TILE - 1u)) == 0u) { /* a tile border: is the next tile this kernel's? */
      t = (job_lo + q) / WGA_TILE;
      if (WGA_UNI32(a.tdesc[t].neg) & WGA_S_SKIP) { /* v1's: finish what is under way, start again behind it */
        if (pos != C_known) {
          fin = true;
          continue;
        }
        live = false;
        t++;
        continue;
      }
    }
    /* room for 256 more events */
    if (e0 > 1u) { /* the FIFO's live entries (from the last gap in front of pos) move to its start */
      const u32 from = e0 - 1u, count = nf + 3u - from;
      for (u32 i0 = 0; i0 < count; i0 += 64u) {
        const u32 i = i0 + lane;
        u64 v = 0;
        if (i < count) v = *(const u64*)(s_fifo + 2u * (from + i));
        WGA_WAVE_SYNC();
        if (i < count) *(u64*)(s_fifo + 2u * i) = v;
        WGA_WAVE_SYNC();
      }
      nf -= from;
      e0 = 1u;
    }
    if (nf + 256u > WGA_S_FIFO) { /* dense gaps: write what is known (partial kilobytes) so that the FIFO drains */
      fin = true;
      if (pos != C_known) continue;
      fin = false;
      if (MODE == WGA_S_PSEUDO) { /* what is left stands on the last column: runs of skip events */
        fifo_dedupe();
        if (nf + 256u > WGA_S_FIFO) break; /* cannot happen: a column holds at most one skip run and the gap events in front of it */
      }
    }
    /* ---- take 256 ops in ---- */
    q_lo = q;
    q_hi = q + 256u < q_end ? q + 256u : q_end;
    q = q_hi;
    ow[0] = own[0], ow[1] = own[1], ow[2] = own[2], ow[3] = own[3];
    buf_load16(obuf, q_hi < q_end ? q_hi * 4u + lane * 16u : WGA_BUF_OOB, own); /* the next 256 travel while these are worked on */
    u32 l[4], g[4], sl = 0, sg = 0, sc = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const bool valid = q_lo + 4u * lane + (u32)e < q_hi;
      const u32 len = valid ? ow[e] >> 4 : 0u, code = ow[e] & 15u;
      l[e] = len & bit_mask(COL_MASK, code);                                           /* columns of the row                          */
      g[e] = (len & bit_mask(GAP_MASK, code)) - (len & bit_mask(SKIP_MASK, code));     /* its gaps (+) and skipped source bytes (-)   */
      sl += l[e];
      sg += g[e];
      sc += g[e] != 0u ? 1u : 0u;
    }
    const u32 il = wave_incl_scan_u32(sl), ig = wave_incl_scan_u32(sg), ic = wave_incl_scan_u32(sc);
    xl0 = C_known + il - sl;
    xg0 = cum_known + ig - sg;
    {
      u32 xl = xl0, xg = xg0, xc = nf + ic - sc;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (g[e] != 0u) {
          s_fifo[2u * xc] = xl;
          s_fifo[2u * xc + 1u] = SYM ? (xg << 1) | (bit_mask(0x404u, ow[e] & 15u) & 1u) : xg;
          xc++;
          xg += g[e];
        }
        xl += l[e];
      }
    }
    C_known += wave_last_u32(il);
    cum_known += wave_last_u32(ig);
    nf += wave_last_u32(ic);
    if (lane < 3u) {
      s_fifo[2u * (nf + lane)] = C_known;
      s_fifo[2u * (nf + lane) + 1u] = SYM ? cum_known << 1 : cum_known;
    }
    WGA_WAVE_SYNC();
    if (re < q_hi) { /* a record ends among these ops */
      const u32 kb = re - q_lo, e = kb & 3u; /* the columns / gap bases in front of op kb: its lane's prefix + its ops in front */
      u32 vl = xl0, vg = xg0;
#pragma unroll
      for (u32 i = 0; i < 3u; i++) {
        const bool valid = q_lo + 4u * lane + i < q_hi && i < e;
        const u32 len = valid ? ow[i] >> 4 : 0u, code = ow[i] & 15u;
        vl += len & bit_mask(COL_MASK, code);
        vg += (len & bit_mask(GAP_MASK, code)) - (len & bit_mask(SKIP_MASK, code));
      }
      C_b = wave_get_u32_dyn(vl, kb >> 2);
      cum_b = wave_get_u32_dyn(vg, kb >> 2);
      bnd = true;
    }
  }
}

/* job -> XCD: blocks go to the 8 XCDs round robin; every XCD works through one contiguous eighth of the jobs (shared output
 * lines and neighbouring source chunks meet in one L2), as v1's tiles do */
__device__ __forceinline__ u64 xcd_job_of_block() {
  const u32 nb = gridDim.x, b = blockIdx.x, x = b & 7u, q = nb >> 3, r = nb & 7u;
  return (u64)(x * q + (x < r ? x : r) + (b >> 3));
}

/* one block = two waves = the target row and the query row of one job of `job_tiles` consecutive tiles */
__device__ __forceinline__ void expand_stream(const ExpandArgs& a) {
  __shared__ u32x4_a16 s_mem[2u * WGA_S_WAVE_BYTES / 16u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 nt = (a.n_ops + WGA_TILE - 1) / WGA_TILE;
  const u64 job = xcd_job_of_block();
  const u64 t0 = job * a.job_tiles;
  if (t0 >= nt) return;
  const u64 t1 = t0 + a.job_tiles < nt ? t0 + a.job_tiles : nt;
  u8* const lds = (u8*)s_mem + wave * WGA_S_WAVE_BYTES;
  if (wave == 0u)
    stream_row<0>(a, lds, lane, t0, t1);
  else
    stream_row<1>(a, lds, lane, t0, t1);
}
/* pafpseudo's rows (K6): every wave of the block walks a job of its own */
template <int MODE>
__device__ __forceinline__ void pseudo_stream(const ExpandArgs& a) {
  __shared__ u32x4_a16 s_mem[2u * WGA_S_WAVE_BYTES / 16u];
  const u32 lane = threadIdx.x & 63u, wave = WGA_WAVE_ID(threadIdx.x);
  const u64 nt = (a.n_ops + WGA_TILE - 1) / WGA_TILE;
  const u64 t0 = (2u * xcd_job_of_block() + wave) * a.job_tiles;
  if (t0 >= nt) return;
  const u64 t1 = t0 + a.job_tiles < nt ? t0 + a.job_tiles : nt;
  stream_row<MODE>(a, (u8*)s_mem + wave * WGA_S_WAVE_BYTES, lane, t0, t1);
}
#ifndef WGA_S_WAVES_PER_SIMD
#define WGA_S_WAVES_PER_SIMD 5 /* launch bound: 96 VGPRs (the natural need is 102: two spill slots), LDS allows 31 waves per CU; 4: 6.0 ms, 5: 5.66 ms, 6 (61 spill slots): 7.4 ms */
#endif
__global__ __launch_bounds__(128, WGA_S_WAVES_PER_SIMD) void k_paf2maf_expand_s(ExpandArgs a) { expand_stream(a); }
__global__ __launch_bounds__(128, WGA_S_WAVES_PER_SIMD) void k_pafpseudo_stream(ExpandArgs a) { pseudo_stream<WGA_S_PSEUDO>(a); }
#ifndef WGA_S_SYM_WAVES
#define WGA_S_SYM_WAVES 5 /* 5: 3.87 ms, 6 (80 VGPRs, 12 spill slots): 3.97, 7: 4.34, 8: 5.80 (configs[1]'s batch) */
#endif
__global__ __launch_bounds__(128, WGA_S_SYM_WAVES) void k_pafpseudo_stream_sym(ExpandArgs a) { pseudo_stream<WGA_S_SYMBOL>(a); }

#endif
