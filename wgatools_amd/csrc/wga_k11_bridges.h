/*
 * wga_k11_bridges.h — K11: bridges between run / data-line lists and packed ops / CIGAR text.
 * One header per kernel family; wga_capi.cpp includes them in dependency order (a header may use helpers of the ones in front of it).
 */
#ifndef WGA_K11_BRIDGES_H
#define WGA_K11_BRIDGES_H

#include "wga_k9_bed.h" /* dec_digits, dec_write, lds_text_flush */

/* ============================================================================================ */
/* K11: bridges between the run / data-line lists and the packed-op and CIGAR-text forms        */
/*      (maf2chain, chain2paf, chain2maf, maf2paf's cg:Z: text)      */
/* ============================================================================================ */
/* All four share one skeleton: element x (a K3 run, or a chain data line) of record r produces
 * src.size(x, r) output units (packed ops or text bytes); an exclusive scan over the elements
 * gives every element its place inside its record's output, which starts at out_off[r].  One
 * thread per element; its record is found by bisection in the CSR offsets. */
__device__ __forceinline__ u32 csr_find_rec(const u64* __restrict__ off, u32 n, u64 x) {
  u32 lo = 0, hi = n; /* largest r < n with off[r] <= x (records without elements are skipped) */
  while (hi - lo > 1u) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (off[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ u32 split_pieces(u64 len) { /* pieces of at most WGA_OP_MAX_LEN, none for 0 */
  return (u32)((len + (u64)WGA_OP_MAX_LEN - 1ull) / (u64)WGA_OP_MAX_LEN);
}
__device__ __forceinline__ u32* put_split(u32* p, u64 len, u32 code, u32 cont) {
  bool first = true;
  while (len) {
    const u64 piece = len > (u64)WGA_OP_MAX_LEN ? (u64)WGA_OP_MAX_LEN : len;
    *p++ = ((u32)piece << 4) | (first ? code : cont);
    len -= piece;
    first = false;
  }
  return p;
}
__device__ __forceinline__ u8* put_len_op(u8* p, u64 len, u8 op) {
  const u32 nd = dec_digits(len);
  dec_write(p, len, nd);
  p[nd] = op;
  return p + nd + 1u;
}

/* K3 runs (start_column << 3 | class, class 0 '=' 1 I 2 D 3 X) of MAF column pairs */
struct MafRunSrc {
  const u64* runs;
  const u64* run_off;
  const u64* cols;
  __device__ u64 len(u64 x, u32 r) const {
    const u64 start = runs[x] >> 3;
    const u64 end = x + 1 < run_off[r + 1] ? runs[x + 1] >> 3 : cols[r];
    return end - start;
  }
  __device__ u32 cls(u64 x) const { return (u32)(runs[x] & 7ull); }
};
/* -> packed ops: '=' 7, I 1, D 2, X 8; a run of 2^28 columns or more is split like the PAF packer
 * splits a length (continuation codes for I / D), so every consumer of a wga_cigar_batch applies */
struct MafRunOps {
  typedef u32 out_t;
  MafRunSrc s;
  __device__ u64 size(u64 x, u32 r) const { return split_pieces(s.len(x, r)); }
  __device__ void write(u64 x, u32 r, u32* p) const {
    const u32 c = s.cls(x);
    const u32 code = c == 0u ? (u32)WGA_OP_EQ : c == 1u ? (u32)WGA_OP_I : c == 2u ? (u32)WGA_OP_D : (u32)WGA_OP_X;
    const u32 cont = c == 1u ? (u32)WGA_OP_I_CONT : c == 2u ? (u32)WGA_OP_D_CONT : code;
    put_split(p, s.len(x, r), code, cont);
  }
};
/* -> the cg:Z: text of maf2paf, "<len><=|I|D|X>" per run (maf.rs:484-520, cigar.rs:400-401) */
struct MafRunText {
  typedef u8 out_t;
  MafRunSrc s;
  __device__ u64 size(u64 x, u32 r) const { return dec_digits(s.len(x, r)) + 1u; }
  __device__ void write(u64 x, u32 r, u8* p) const {
    const u32 c = s.cls(x);
    put_len_op(p, s.len(x, r), c == 0u ? (u8)'=' : c == 1u ? (u8)'I' : c == 2u ? (u8)'D' : (u8)'X');
  }
};
/* chain data lines, three u64 each: size, 2nd column (bases only in the target: D), 3rd column
 * (bases only in the query: I) — chain.rs:330-348 reads them in this order */
struct ChainLineSrc {
  const u64* lines;
  __device__ u64 size_(u64 x) const { return lines[3 * x]; }
  __device__ u64 del_(u64 x) const { return lines[3 * x + 1]; }
  __device__ u64 ins_(u64 x) const { return lines[3 * x + 2]; }
};
/* -> packed ops in the order parse_chain_to_cigar / parse_chain_to_insert walk a line
 * (cigar.rs:576-606, converter.rs:360-388): M size, I 3rd column, D 2nd column; zero lengths
 * have no effect on rows or counts and are left out */
struct ChainLineOps {
  typedef u32 out_t;
  ChainLineSrc s;
  __device__ u64 size(u64 x, u32) const {
    return (u64)split_pieces(s.size_(x)) + split_pieces(s.ins_(x)) + split_pieces(s.del_(x));
  }
  __device__ void write(u64 x, u32, u32* p) const {
    p = put_split(p, s.size_(x), (u32)WGA_OP_M, (u32)WGA_OP_M);
    p = put_split(p, s.ins_(x), (u32)WGA_OP_I, (u32)WGA_OP_I_CONT);
    put_split(p, s.del_(x), (u32)WGA_OP_D, (u32)WGA_OP_D_CONT);
  }
};
/* -> chain2paf's CIGAR text: "<size>M" always, "<n>I" / "<n>D" when non-zero (cigar.rs:576-606) */
struct ChainLineText {
  typedef u8 out_t;
  ChainLineSrc s;
  __device__ u64 size(u64 x, u32) const {
    const u64 i = s.ins_(x), d = s.del_(x);
    return (u64)dec_digits(s.size_(x)) + 1u + (i ? dec_digits(i) + 1u : 0u) + (d ? dec_digits(d) + 1u : 0u);
  }
  __device__ void write(u64 x, u32, u8* p) const {
    const u64 i = s.ins_(x), d = s.del_(x);
    p = put_len_op(p, s.size_(x), (u8)'M');
    if (i) p = put_len_op(p, i, (u8)'I');
    if (d) put_len_op(p, d, (u8)'D');
  }
};

template <typename F>
struct ScanElem { /* scan functor: output units of element x */
  F f;
  const u64* elem_off;
  u32 n;
  __device__ u64 operator()(u32 x) const { return f.size((u64)x, csr_find_rec(elem_off, n, (u64)x)); }
};
__global__ __launch_bounds__(256) void k_elem_rec_totals(u32 n, const u64* __restrict__ elem_off,
                                                         const u64* __restrict__ esc, u64* __restrict__ cnt) {
  const u32 r = blockIdx.x * 256u + threadIdx.x;
  if (r < n) cnt[r] = esc[elem_off[r + 1]] - esc[elem_off[r]];
}
/* What a block of the fill needs to know before it can start, found once per block by a small kernel in front of the fill
 * (one thread per block, two bisections): the records of its first and last element and, in units of the output, where the
 * block's first unit goes when the block lies inside one record.  The fill's blocks used to search for their records themselves
 * — two wave-wide searches of three dependent probes each, then out_off[r] and esc[elem_off[r]] behind them: five dependent
 * round trips in front of the first useful load of a block that lives for a few microseconds (round 6). */
struct __attribute__((aligned(16))) wga_elem_block {
  u32 r_lo, r_hi;
  u64 base; /* out_off[r_lo] + (units of r_lo's elements in front of the block) */
};
__global__ __launch_bounds__(256) void k_elem_blocks(u32 n, u32 ne, const u64* __restrict__ elem_off, const u64* __restrict__ esc,
                                                     const u64* __restrict__ out_off, wga_elem_block* __restrict__ blocks) {
  const u32 b = blockIdx.x * 256u + threadIdx.x;
  const u64 x0 = (u64)b * 256u;
  if (x0 >= (u64)ne) return;
  const u64 x1 = x0 + 256u < (u64)ne ? x0 + 256u : (u64)ne;
  wga_elem_block e;
  e.r_lo = csr_find_rec(elem_off, n, x0);
  e.r_hi = csr_find_rec(elem_off, n, x1 - 1u);
  e.base = out_off[e.r_lo] + (esc[x0] - esc[elem_off[e.r_lo]]);
  blocks[b] = e;
}
/* One thread per element, 256 consecutive elements per block (k_elem_blocks tells the block where it stands).
 * A block whose elements belong to ONE record writes one contiguous stretch of that record's output: its threads put
 * their units into an LDS buffer that mirrors the stretch's position inside its 16-byte group, and the stretch goes out
 * in 16-byte stores.  Blocks across a record border, or with more output than the buffer holds, write directly. */
#define WGA_ELEM_STAGE 16384u
template <typename F>
__global__ __launch_bounds__(256) void k_elem_fill(F f, u32 n, u32 ne, const u64* __restrict__ elem_off,
                                                   const u64* __restrict__ esc, typename F::out_t* out,
                                                   const u64* __restrict__ out_off, const wga_elem_block* __restrict__ blocks) {
  typedef typename F::out_t out_t;
  __shared__ u32x4_a16 s_buf[(WGA_ELEM_STAGE + 32u) / 16u];
  const u32 tid = threadIdx.x;
  const u32 x0 = blockIdx.x * 256u, x1 = x0 + 256u < ne ? x0 + 256u : ne;
  const wga_elem_block eb = blocks[blockIdx.x];
  const u32 r_lo = eb.r_lo, r_hi = eb.r_hi;
  const u32 x = x0 + tid;
  const u64 e0 = esc[x0], e1 = esc[x1]; /* units in front of the block, and behind it */
  const bool staged = r_lo == r_hi && (e1 - e0) * sizeof(out_t) <= (u64)WGA_ELEM_STAGE;
  if (staged) { /* block-uniform */
    out_t* const g0 = out + eb.base;
    const u32 a = (u32)((uintptr_t)g0 & 15u);
    u8* const tbuf = (u8*)s_buf;
    if (x < x1) f.write((u64)x, r_lo, (out_t*)(tbuf + a) + (esc[x] - e0));
    __syncthreads();
    lds_text_flush(tbuf, a, (u32)((e1 - e0) * sizeof(out_t)), (u8*)g0 - a, tid, 256u);
    return;
  }
  if (x >= x1) return;
  u32 lo = r_lo, hi = r_hi + 1u; /* largest r in [r_lo, r_hi] with elem_off[r] <= x */
  while (hi - lo > 1u) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (elem_off[mid] <= (u64)x) lo = mid; else hi = mid;
  }
  f.write((u64)x, lo, out + out_off[lo] + (esc[x] - esc[elem_off[lo]]));
}

#endif /* WGA_K11_BRIDGES_H */
