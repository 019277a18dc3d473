/*
 * wga_k6_pafpseudo.h — K6: pafpseudo — target-coordinate pseudo-MAF segments, block kernel (gen_pesudo_maf_by_cigar, cigar.rs:744-804).
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K6_PAFPSEUDO_H
#define WGA_K6_PAFPSEUDO_H

#include "wga_kernels.h"

/* ============================================================================================ */
/* K6: pafpseudo                                                                                */
/* ============================================================================================ */
struct PseudoArgs {
  const u32* ops;
  const u64* op_off;
  const u8* strand_neg;
  u32 n;
  u64 n_ops;
  const wga_tile_sum* tiles;
  const wga_class_sums* rec_sums;
  int base_mode;
  const u8* q_fa;
  u64 q_fa_bytes;
  const u64* q_src_off;
  const u64* q_src_len;
  const u64* skip;
  u8* out;
  const u64* dst_off;
  wga_rec_diag* diag;
  const u32* tile_count; /* k_pafpseudo_fill_list: the blocks loop over tile_list[0 .. *tile_count) */
  const u32* tile_list;
};

/* symbol mode: '1' for M/=, '0' for X, '-' for D, nothing for the rest (cigar.rs:760-796) */
__device__ __forceinline__ u32 pseudo_symbol(u32 code) {
  return (code == WGA_OP_M || code == WGA_OP_EQ)
             ? 0x31313131u
             : code == WGA_OP_X ? 0x30303030u
                                : (code == WGA_OP_D || code == WGA_OP_D_CONT) ? 0x2D2D2D2Du : 0u;
}

/* fill N bytes whose value depends only on the op covering the column (symbol mode) */
__device__ __forceinline__ void emit_symbols(u8* dst, u32 N, u32 c0, const u32* s_col,
                                             const u32* s_sym, int ka, int kb, const u32* optbl, u32 gsh,
                                             const u32x4_a16* lowmask, u32 tid, u32 nthreads) {
  if (N == 0) return;
  const u64 A = (u64)dst, E = A + N;
  const u64 first = A >> 4, last = (E - 1) >> 4;
  for (u64 ch = first + tid; ch <= last; ch += nthreads) {
    const u64 base_addr = ch << 4;
    const u32 a0 = base_addr < A ? (u32)(A - base_addr) : 0u;
    const u32 b0 = base_addr + 16 > E ? (u32)(E - base_addr) : 16u;
    const u32 cz = c0 + (u32)(base_addr - A);
    u32 c = cz + a0;
    const u32 c_end = cz + b0;
    /* an op that starts at or before c: the last one that starts before c's granule (optbl = ops that start
     * before each granule), or the segment's first; ops that end before c are stepped over below */
    (void)kb;
    int k = (int)optbl[c >> gsh] - 1; /* c, not cz: cz wraps below zero for a row that starts mid-granule */
    k = k < ka ? ka : k;
    u32 o[4] = {0u, 0u, 0u, 0u};
    while (c < c_end) {
      u32 oe = s_col[k + 1];
      u32 pe = oe < c_end ? oe : c_end;
      if (pe > c) {
        const u32 sym = s_sym[k];
        const u32 W[4] = {sym, sym, sym, sym};
        merge16(o, W, (int)(c - cz), (int)(pe - cz), lowmask);
        c = pe;
      }
      k++;
    }
    if (a0 == 0u && b0 == 16u) {
      u32x4_a16 v = {o[0], o[1], o[2], o[3]};
      *(u32x4_a16*)base_addr = v;
    } else {
      u8* p = (u8*)base_addr;
      for (u32 j = a0; j < b0; j++) {
        u32 d = j >> 2;
        u32 word = d == 0 ? o[0] : d == 1 ? o[1] : d == 2 ? o[2] : o[3];
        p[j] = (u8)(word >> (8u * (j & 3u)));
      }
    }
  }
}

/* BASE = base mode (query bases; the row emitter of K2) or symbol mode: two kernels, so that the symbol one does
 * not carry the emitter's registers and LDS */
#ifndef WGA_K6_BLOCKS_BASE
#define WGA_K6_BLOCKS_BASE 4
#endif
#ifndef WGA_K6_BLOCKS_SYM
#define WGA_K6_BLOCKS_SYM 6
#endif
template <bool BASE>
__device__ __forceinline__ void pseudo_tile(const PseudoArgs& a, const u64 g) {
  /* the event lists and the chunk queue belong to the row emitter (base mode); symbol mode keeps 13 KB of LDS */
  __shared__ u32 s_col[WGA_TILE + 1];                  /* exclusive prefix of target columns (M = X D)          */
  __shared__ u32 s_ev[WGA_TILE + 1];                   /* exclusive count of event ops (D, I, S)                */
  __shared__ u32 s_sym[BASE ? 1 : WGA_TILE + 1];       /* symbol-mode byte of the op                            */
  __shared__ u32 s_g_col[BASE ? WGA_TILE + 2 : 2];     /* events: column                                        */
  __shared__ u32 s_g_cum[BASE ? WGA_TILE + 2 : 2];     /*         '-' bases before (D)                          */
  __shared__ u32 s_g_adj[BASE ? WGA_TILE + 2 : 2];     /*         D bases - (I+S) bases before (wrapping)       */
  __shared__ u32 s_tbl[WGA_TBL_N + 2];                 /* events that start before each column granule          */
  __shared__ u32 s_zero2[2];
  __shared__ u64 s_w[5];
  __shared__ u32 s_w4[4];
  __shared__ u64 s_red[4][4];
  __shared__ u32x4_a16 s_lowmask[17];
  __shared__ u32 s_queue[BASE ? 4 * WGA_QCAP : 4];

  const u32 tid = threadIdx.x;
  build_lowmask(s_lowmask);
  const u32 lane = tid & 63u, wave = WGA_WAVE_ID(tid);
  const u64 tile_start = g * WGA_TILE;
  const u64 tile_end = tile_start + WGA_TILE < a.n_ops ? tile_start + WGA_TILE : a.n_ops;
  const u32 nt = (u32)(tile_end - tile_start);
  const wga_tile_sum tsum = a.tiles[g];
  const bool fast = tsum.tot[CLS_MX] + tsum.tot[CLS_D] + tsum.tot[CLS_I] + tsum.tot[CLS_S] <=
                    WGA_FAST_COL_LIMIT;
  u32 gsh = WGA_TBL_SHIFT;
  while (((tsum.tot[CLS_MX] + tsum.tot[CLS_D]) >> gsh) >= WGA_TBL_N) gsh++;
  if (fast)
    for (u32 k = tid; k < WGA_TBL_N + 2u; k += WGA_BLOCK) s_tbl[k] = 0u;
  if (tid < 2u) s_zero2[tid] = 0u;

  u32 opw[4];
  {
    u32 base = tid * 4u;
    if (base + 3 < nt) {
      u32x4_a16 v = *(const u32x4_a16*)(a.ops + tile_start + base);
      opw[0] = v[0];
      opw[1] = v[1];
      opw[2] = v[2];
      opw[3] = v[3];
    } else {
      for (int e = 0; e < 4; e++) opw[e] = (base + e < nt) ? a.ops[tile_start + base + e] : 0u;
    }
  }
  if (fast) {
    u32 cls[4];
    u32 l[4], sl = 0, sd = 0, sis = 0, cnt = 0;
    for (int e = 0; e < 4; e++) {
      u32 len = opw[e] >> 4;
      cls[e] = op_class(opw[e] & 15u);
      l[e] = (cls[e] == CLS_MX || cls[e] == CLS_D) ? len : 0u;
      sl += l[e];
      sd += cls[e] == CLS_D ? len : 0u;
      sis += (cls[e] == CLS_I || cls[e] == CLS_S) ? len : 0u;
      cnt += (cls[e] == CLS_D || cls[e] == CLS_I || cls[e] == CLS_S) ? 1u : 0u;
    }
    u64 totA, totB;
    u64 exA = block_excl_scan_u64((u64)sl | ((u64)sd << 32), s_w, &totA);
    u64 exB = block_excl_scan_u64((u64)sis | ((u64)cnt << 32), s_w, &totB);
    u32 x_col = (u32)exA, x_d = (u32)(exA >> 32), x_is = (u32)exB, x_cnt = (u32)(exB >> 32);
    for (int e = 0; e < 4; e++) {
      u32 k = tid * 4u + (u32)e;
      s_col[k] = x_col;
      s_ev[k] = x_cnt;
      if (!BASE) {
        s_sym[k] = pseudo_symbol(opw[e] & 15u);
        if (k < nt) atomicAdd(&s_tbl[x_col >> gsh], 1u); /* symbol mode: the table counts op starts per granule */
      }
      if (cls[e] == CLS_D || cls[e] == CLS_I || cls[e] == CLS_S) {
        if (BASE) {
          s_g_col[x_cnt] = x_col;
          s_g_cum[x_cnt] = x_d;
          s_g_adj[x_cnt] = x_d - x_is;
          tbl_mark_event(s_tbl, x_col, cls[e] == CLS_D ? (opw[e] >> 4) : 0u, gsh, 0u);
        }
        if (cls[e] == CLS_D)
          x_d += opw[e] >> 4;
        else
          x_is += opw[e] >> 4;
        x_cnt += 1u;
      }
      x_col += l[e];
    }
    if (tid == WGA_BLOCK - 1) {
      s_col[WGA_TILE] = x_col;
      s_ev[WGA_TILE] = x_cnt;
      if (BASE) {
        s_g_col[x_cnt] = s_g_col[x_cnt + 1u] = x_col;
        s_g_cum[x_cnt] = s_g_cum[x_cnt + 1u] = x_d;
        s_g_adj[x_cnt] = s_g_adj[x_cnt + 1u] = x_d - x_is;
      }
    }
    __syncthreads(); /* raw marks -> exclusive prefix */
    tbl_scan(s_tbl, s_w4);
  }
  __syncthreads();

  u32 r = (u32)tsum.rec;
  u64 cur = tile_start;
  while (cur < tile_end) {
    u64 re = a.op_off[r + 1];
    while (re <= cur) {
      r++;
      re = a.op_off[r + 1];
    }
    const u64 rs = a.op_off[r];
    const u64 seg_end = re < tile_end ? re : tile_end;
    const u32 ka = (u32)(cur - tile_start), kb = (u32)(seg_end - tile_start);

    u64 b_mx = 0, b_i = 0, b_d = 0, b_s = 0;
    if (rs < tile_start) {
      const u64 g0 = rs / WGA_TILE;
      u64 p_mx = 0, p_i = 0, p_d = 0, p_s = 0;
      for (u64 k = g0 + tid; k < g; k += WGA_BLOCK) {
        const wga_tile_sum* t = a.tiles + k;
        const u64* v = (k == g0) ? t->tail : t->tot;
        p_mx += v[CLS_MX];
        p_i += v[CLS_I];
        p_d += v[CLS_D];
        p_s += v[CLS_S];
      }
      p_mx = wave_sum_u64(p_mx);
      p_i = wave_sum_u64(p_i);
      p_d = wave_sum_u64(p_d);
      p_s = wave_sum_u64(p_s);
      __syncthreads();
      if (lane == 0) {
        s_red[wave][0] = p_mx;
        s_red[wave][1] = p_i;
        s_red[wave][2] = p_d;
        s_red[wave][3] = p_s;
      }
      __syncthreads();
      for (int w2 = 0; w2 < 4; w2++) {
        b_mx += WGA_UNI64(s_red[w2][0]);
        b_i += WGA_UNI64(s_red[w2][1]);
        b_d += WGA_UNI64(s_red[w2][2]);
        b_s += WGA_UNI64(s_red[w2][3]);
      }
    }
    const u64 cb = b_mx + b_d;       /* target columns of this record before the segment */
    const u64 qb = b_mx + b_i + b_s; /* query bases consumed before it                   */

    const wga_class_sums cs = a.rec_sums[r];
    const u64 T_total = cs.mx + cs.d;          /* columns the CIGAR emits */
    const u64 Q_total = cs.mx + cs.i + cs.s;   /* query bases it consumes */
    RowSrc qs;
    qs.fa = a.q_fa;
    qs.fa_bytes = a.q_fa_bytes;
    qs.src_off = BASE ? a.q_src_off[r] : 0;
    qs.src_len = BASE ? a.q_src_len[r] : 0;
    qs.rc = a.strand_neg[r] != 0;
    /* edited length: String::drain / insert_str semantics (cigar.rs:769-786) */
    /* a record whose I / S ops take more than the slice holds (String::drain panics, reported below) has no row: the
     * difference must not wrap into a row of 2^64 bytes */
    const u64 row_len = BASE ? (qs.src_len + cs.d >= cs.i + cs.s ? qs.src_len + cs.d - (cs.i + cs.s) : 0ull) : T_total;
    const u64 skip = a.skip[r];
    u8* const dst = a.out + a.dst_off[r];
    u64* const bad_base = (u64*)&a.diag[r].bad_base_pos;
    u64* const panic_idx = (u64*)&a.diag[r].panic_op_idx;

    if (fast) {
      /* the same in every lane, but read from LDS: told to the compiler, or the whole row emitter below sits in
       * exec-masked control flow with its loop bounds in VGPRs */
      const u32 col_a = WGA_UNI32(s_col[ka]), seg_cols = WGA_UNI32(s_col[kb]) - col_a;
      const int ea = (int)WGA_UNI32(s_ev[ka]), eb = (int)WGA_UNI32(s_ev[kb]);
      const u32 adj_a = BASE ? WGA_UNI32(s_g_adj[ea]) : 0u;
      if (BASE) {
        /* drain(offset..offset+len) panics past the end of the string, insert_str(offset)
         * beyond it (cigar.rs:772,779): in slice terms, an I/S op needs q_before + len <= slice
         * length, a D op q_before <= slice length */
        const u32 d_a = WGA_UNI32(s_g_cum[ea]), is_a = d_a - adj_a;
        for (int e = 0; e < 4; e++) {
          u32 k = tid * 4u + (u32)e;
          if (k >= ka && k < kb) {
            /* the op is read again (it is not kept in registers across the row emitter), its own prefixes sit in
             * the event lists at its slot (s_ev = events before op k) */
            const u32 op = a.ops[tile_start + k];
            const u32 c = op_class(op & 15u);
            if (c == CLS_I || c == CLS_S || c == CLS_D) {
              const u32 ev = s_ev[k];
              const u32 e_d = s_g_cum[ev], e_is = e_d - s_g_adj[ev];
              u64 q_before = qb + (u64)(s_col[k] - col_a) - (u64)(e_d - d_a) + (u64)(e_is - is_a);
              u64 len = op >> 4;
              if (c != CLS_D && q_before + len > qs.src_len) atomicMin(panic_idx, tile_start + k - rs);
              if (c == CLS_D && q_before > qs.src_len) atomicMin(panic_idx, tile_start + k - rs);
            }
          }
        }
      }
      u64 x0 = cb > skip ? cb : skip;
      u64 x1 = cb + seg_cols < row_len ? cb + seg_cols : row_len;
      if (x1 > x0) {
        const u32 c_first = col_a + (u32)(x0 - cb);
        if (BASE) {
          RowDesc rd;
          rd.c_org = col_a;
          rd.G_col = s_g_col;
          rd.G_cum = s_g_cum;
          rd.G_adj = s_g_adj;
          rd.ga = ea;
          rd.gb = eb;
          rd.gcum_a = adj_a;
          rd.sbase = qb;
          rd.lowmask = s_lowmask;
          rd.tbl = s_tbl;
          rd.tsh = 0u;
          rd.gsh = gsh;
          rd.queue = s_queue;
          rowsrc_prepare(qs, qb);
          emit_row(dst + (x0 - skip), (u32)(x1 - x0), c_first, rd, qs, tid, WGA_BLOCK, bad_base);
        } else
          emit_symbols(dst + (x0 - skip), (u32)(x1 - x0), c_first, s_col, s_sym, (int)ka, (int)kb, s_tbl, gsh,
                       s_lowmask, tid, WGA_BLOCK);
      }
    }
    /* u64 fallback for tiles too wide for u32 columns: op-serial walk, every thread redundantly */
    if (!fast) {
      u64 x = cb, qp = qb;
      for (u64 k = cur; k < seg_end; k++) {
        const u32 op = a.ops[k];
        const u32 code = op & 15u;
        const u32 c = op_class(code);
        const u64 len = op >> 4;
        if (BASE && tid == 0) {
          if ((c == CLS_I || c == CLS_S) && qp + len > qs.src_len) atomicMin(panic_idx, k - rs);
          if (c == CLS_D && qp > qs.src_len) atomicMin(panic_idx, k - rs);
        }
        if (c == CLS_MX || c == CLS_D) {
          for (u64 j = tid; j < len; j += WGA_BLOCK) {
            u64 xx = x + j;
            if (xx >= skip && xx < row_len) {
              u8 v;
              if (BASE)
                v = (c == CLS_D) ? (u8)'-' : src_byte(qs, qp + j, bad_base);
              else
                v = (u8)(pseudo_symbol(code) & 0xFFu);
              dst[xx - skip] = v;
            }
          }
          x += len;
        }
        if (c == CLS_MX || c == CLS_I || c == CLS_S) qp += len;
      }
    }
    /* leftover query bases beyond the CIGAR stay at the end of the edited string */
    if (seg_end == re && BASE && row_len > T_total) {
      u64 x0 = T_total > skip ? T_total : skip;
      if (row_len > x0)
        emit_tail(dst + (x0 - skip), row_len - x0, Q_total + (x0 - T_total), qs, s_lowmask, s_queue, s_zero2, s_g_col, tid, WGA_BLOCK, bad_base);
    }
    cur = seg_end;
    r++;
  }
}
/* one block per tile of the batch: when the streaming row kernel is switched off ("pseudo_variant" 0) */
template <bool BASE>
__global__ __launch_bounds__(256, BASE ? WGA_K6_BLOCKS_BASE : WGA_K6_BLOCKS_SYM) void k_pafpseudo_fill(PseudoArgs a) {
  pseudo_tile<BASE>(a, xcd_tile_of_block());
}
/* the tiles the streaming row kernel (k_pafpseudo_stream / _sym, wga_kernels_k2s.h) leaves: giant tiles and, in base mode, records
 * whose slice is not exactly what their CIGAR consumes (leftover bases, drain / insert_str panics) and slices at a pool's edge */
template <bool BASE>
__global__ __launch_bounds__(256, BASE ? WGA_K6_BLOCKS_BASE : WGA_K6_BLOCKS_SYM) void k_pafpseudo_fill_list(PseudoArgs a) {
  const u32 n_list = *a.tile_count;
  for (u32 idx = blockIdx.x; idx < n_list; idx += gridDim.x) {
    pseudo_tile<BASE>(a, a.tile_list[idx]);
    __syncthreads(); /* the tile's LDS state is dead */
  }
}

#endif /* WGA_K6_PAFPSEUDO_H */
