/*
 * abi_layout.c — the layout of every struct that crosses the C-ABI of libwgahip.so (include/wga_hip.h), frozen as
 * compile-time assertions.  A binding in another language (the Rust `#[repr(C)]` structs of INTEGRATION.md section 2, which
 * have never met a Rust compiler here) must reproduce exactly these sizes and offsets (x86-64 SysV, the only host of an MI355X).
 * tests/test_abi.py compiles this file; a change of include/wga_hip.h that moves a field fails there first.
 * Generated from the header with offsetof / sizeof (scripts: none needed — gcc -Iinclude on a printing twin of this file).
 */
#include <stddef.h>
#include "wga_hip.h"

_Static_assert(WGA_ABI_VERSION == 3, "ABI version");
_Static_assert(sizeof(wga_rec_diag) == 24, "wga_rec_diag");
_Static_assert(offsetof(wga_rec_diag, bad_op_idx) == 0, "wga_rec_diag.bad_op_idx");
_Static_assert(offsetof(wga_rec_diag, panic_op_idx) == 8, "wga_rec_diag.panic_op_idx");
_Static_assert(offsetof(wga_rec_diag, bad_base_pos) == 16, "wga_rec_diag.bad_base_pos");
_Static_assert(sizeof(wga_cigar_counts) == 88, "wga_cigar_counts");
_Static_assert(offsetof(wga_cigar_counts, match) == 0, "wga_cigar_counts.match");
_Static_assert(offsetof(wga_cigar_counts, mismatch) == 8, "wga_cigar_counts.mismatch");
_Static_assert(offsetof(wga_cigar_counts, ins_ev) == 16, "wga_cigar_counts.ins_ev");
_Static_assert(offsetof(wga_cigar_counts, ins_bp) == 24, "wga_cigar_counts.ins_bp");
_Static_assert(offsetof(wga_cigar_counts, del_ev) == 32, "wga_cigar_counts.del_ev");
_Static_assert(offsetof(wga_cigar_counts, del_bp) == 40, "wga_cigar_counts.del_bp");
_Static_assert(offsetof(wga_cigar_counts, inv_ins_ev) == 48, "wga_cigar_counts.inv_ins_ev");
_Static_assert(offsetof(wga_cigar_counts, inv_ins_bp) == 56, "wga_cigar_counts.inv_ins_bp");
_Static_assert(offsetof(wga_cigar_counts, inv_del_ev) == 64, "wga_cigar_counts.inv_del_ev");
_Static_assert(offsetof(wga_cigar_counts, inv_del_bp) == 72, "wga_cigar_counts.inv_del_bp");
_Static_assert(offsetof(wga_cigar_counts, inv_ev) == 80, "wga_cigar_counts.inv_ev");
_Static_assert(sizeof(wga_cigar_batch) == 40, "wga_cigar_batch");
_Static_assert(offsetof(wga_cigar_batch, d_ops) == 0, "wga_cigar_batch.d_ops");
_Static_assert(offsetof(wga_cigar_batch, d_op_off) == 8, "wga_cigar_batch.d_op_off");
_Static_assert(offsetof(wga_cigar_batch, d_strand_neg) == 16, "wga_cigar_batch.d_strand_neg");
_Static_assert(offsetof(wga_cigar_batch, n_ops) == 24, "wga_cigar_batch.n_ops");
_Static_assert(offsetof(wga_cigar_batch, n) == 32, "wga_cigar_batch.n");
_Static_assert(sizeof(wga_tok_err) == 16, "wga_tok_err");
_Static_assert(offsetof(wga_tok_err, err) == 0, "wga_tok_err.err");
_Static_assert(offsetof(wga_tok_err, tok_len) == 4, "wga_tok_err.tok_len");
_Static_assert(offsetof(wga_tok_err, tok_off) == 8, "wga_tok_err.tok_off");
_Static_assert(sizeof(wga_vcf_rec) == 88, "wga_vcf_rec");
_Static_assert(offsetof(wga_vcf_rec, t_name_off) == 0, "wga_vcf_rec.t_name_off");
_Static_assert(offsetof(wga_vcf_rec, q_name_off) == 8, "wga_vcf_rec.q_name_off");
_Static_assert(offsetof(wga_vcf_rec, t_name_len) == 16, "wga_vcf_rec.t_name_len");
_Static_assert(offsetof(wga_vcf_rec, q_name_len) == 20, "wga_vcf_rec.q_name_len");
_Static_assert(offsetof(wga_vcf_rec, t_start) == 24, "wga_vcf_rec.t_start");
_Static_assert(offsetof(wga_vcf_rec, t_end) == 32, "wga_vcf_rec.t_end");
_Static_assert(offsetof(wga_vcf_rec, q_start) == 40, "wga_vcf_rec.q_start");
_Static_assert(offsetof(wga_vcf_rec, q_end) == 48, "wga_vcf_rec.q_end");
_Static_assert(offsetof(wga_vcf_rec, t_off) == 56, "wga_vcf_rec.t_off");
_Static_assert(offsetof(wga_vcf_rec, t_len) == 64, "wga_vcf_rec.t_len");
_Static_assert(offsetof(wga_vcf_rec, q_off) == 72, "wga_vcf_rec.q_off");
_Static_assert(offsetof(wga_vcf_rec, q_len) == 80, "wga_vcf_rec.q_len");
_Static_assert(sizeof(wga_maf_vcf_rec) == 56, "wga_maf_vcf_rec");
_Static_assert(offsetof(wga_maf_vcf_rec, t_name_off) == 0, "wga_maf_vcf_rec.t_name_off");
_Static_assert(offsetof(wga_maf_vcf_rec, q_name_off) == 8, "wga_maf_vcf_rec.q_name_off");
_Static_assert(offsetof(wga_maf_vcf_rec, t_name_len) == 16, "wga_maf_vcf_rec.t_name_len");
_Static_assert(offsetof(wga_maf_vcf_rec, q_name_len) == 20, "wga_maf_vcf_rec.q_name_len");
_Static_assert(offsetof(wga_maf_vcf_rec, t_start) == 24, "wga_maf_vcf_rec.t_start");
_Static_assert(offsetof(wga_maf_vcf_rec, q_start) == 32, "wga_maf_vcf_rec.q_start");
_Static_assert(offsetof(wga_maf_vcf_rec, q_size) == 40, "wga_maf_vcf_rec.q_size");
_Static_assert(offsetof(wga_maf_vcf_rec, q_neg) == 48, "wga_maf_vcf_rec.q_neg");
_Static_assert(sizeof(wga_vcf_err) == 16, "wga_vcf_err");
_Static_assert(offsetof(wga_vcf_err, item) == 0, "wga_vcf_err.item");
_Static_assert(offsetof(wga_vcf_err, kind) == 8, "wga_vcf_err.kind");
_Static_assert(offsetof(wga_vcf_err, ch) == 12, "wga_vcf_err.ch");
_Static_assert(sizeof(wga_bgzf_block) == 24, "wga_bgzf_block");
_Static_assert(offsetof(wga_bgzf_block, in_off) == 0, "wga_bgzf_block.in_off");
_Static_assert(offsetof(wga_bgzf_block, in_len) == 8, "wga_bgzf_block.in_len");
_Static_assert(offsetof(wga_bgzf_block, out_len) == 12, "wga_bgzf_block.out_len");
_Static_assert(offsetof(wga_bgzf_block, out_off) == 16, "wga_bgzf_block.out_off");
_Static_assert(sizeof(wga_chain_trim_t) == 32, "wga_chain_trim_t");
_Static_assert(offsetof(wga_chain_trim_t, head_ins) == 0, "wga_chain_trim_t.head_ins");
_Static_assert(offsetof(wga_chain_trim_t, head_del) == 8, "wga_chain_trim_t.head_del");
_Static_assert(offsetof(wga_chain_trim_t, tail_ins) == 16, "wga_chain_trim_t.tail_ins");
_Static_assert(offsetof(wga_chain_trim_t, tail_del) == 24, "wga_chain_trim_t.tail_del");
_Static_assert(sizeof(wga_paf_line) == 120, "wga_paf_line");
_Static_assert(offsetof(wga_paf_line, num) == 0, "wga_paf_line.num");
_Static_assert(offsetof(wga_paf_line, qname_off) == 72, "wga_paf_line.qname_off");
_Static_assert(offsetof(wga_paf_line, tname_off) == 80, "wga_paf_line.tname_off");
_Static_assert(offsetof(wga_paf_line, cg_beg) == 88, "wga_paf_line.cg_beg");
_Static_assert(offsetof(wga_paf_line, cg_end) == 96, "wga_paf_line.cg_end");
_Static_assert(offsetof(wga_paf_line, qname_len) == 104, "wga_paf_line.qname_len");
_Static_assert(offsetof(wga_paf_line, tname_len) == 108, "wga_paf_line.tname_len");
_Static_assert(offsetof(wga_paf_line, n_fields) == 112, "wga_paf_line.n_fields");
_Static_assert(offsetof(wga_paf_line, strand_neg) == 116, "wga_paf_line.strand_neg");
_Static_assert(offsetof(wga_paf_line, status) == 117, "wga_paf_line.status");
_Static_assert(sizeof(wga_maf_line) == 56, "wga_maf_line");
_Static_assert(offsetof(wga_maf_line, num) == 0, "wga_maf_line.num");
_Static_assert(offsetof(wga_maf_line, name_off) == 24, "wga_maf_line.name_off");
_Static_assert(offsetof(wga_maf_line, seq_off) == 32, "wga_maf_line.seq_off");
_Static_assert(offsetof(wga_maf_line, seq_len) == 40, "wga_maf_line.seq_len");
_Static_assert(offsetof(wga_maf_line, name_len) == 48, "wga_maf_line.name_len");
_Static_assert(offsetof(wga_maf_line, strand_neg) == 52, "wga_maf_line.strand_neg");
_Static_assert(offsetof(wga_maf_line, status) == 53, "wga_maf_line.status");
_Static_assert(sizeof(wga_fa_contig) == 32, "wga_fa_contig");
_Static_assert(offsetof(wga_fa_contig, hdr_start) == 0, "wga_fa_contig.hdr_start");
_Static_assert(offsetof(wga_fa_contig, hdr_end) == 8, "wga_fa_contig.hdr_end");
_Static_assert(offsetof(wga_fa_contig, pool_off) == 16, "wga_fa_contig.pool_off");
_Static_assert(offsetof(wga_fa_contig, len) == 24, "wga_fa_contig.len");
_Static_assert(sizeof(wga_class_sums) == 40, "wga_class_sums");
_Static_assert(offsetof(wga_class_sums, mx) == 0, "wga_class_sums.mx");
_Static_assert(offsetof(wga_class_sums, i) == 8, "wga_class_sums.i");
_Static_assert(offsetof(wga_class_sums, d) == 16, "wga_class_sums.d");
_Static_assert(offsetof(wga_class_sums, s) == 24, "wga_class_sums.s");
_Static_assert(offsetof(wga_class_sums, o) == 32, "wga_class_sums.o");
int wga_abi_layout_checked(void) { return WGA_ABI_VERSION; }
