/*
 * oracle.h — CPU restatement of wgatools' CIGAR hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library; it
 * is the checker, never the product path (the product is libwgahip.so and fails loudly without a
 * GPU).  Every function restates one reference function, deliberately naive (same algorithmic
 * structure, including the quadratic String::insert_str / drain behaviour), and cites the
 * /root/reference-relative file:line it follows.
 *
 * PARITY PINNING.  The reference (Rust) cannot be built or run in this environment and has no
 * unit tests.  Pinned: orc_call_within_var + orc_find_safe_chunk_boundary + orc_cigar_cat_ext_caller
 * reproduce the golden VCF of the reference repository (README.md:323-343, `wgatools call
 * test/test.maf -s -l0`), and orc_cigar_to_base_plotdata reproduces the data rows of its committed
 * test/test.html (`dotplot` of test/testdotplot.paf, record 1); both in tests/test_oracle_golden.py.
 * Everything else — stat, paf2maf, maf2paf, pafcov, pafpseudo, the chain converters — is checked
 * only against expected outputs derived by reading the code: for those
 * paths this oracle is **parity unpinned**.
 */
#ifndef WGA_ORACLE_H
#define WGA_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* WGAError variants reachable from the hot path (errors.rs:45-74) + PANIC for the spots where
 * the reference panics instead of returning Err. */
enum orc_err_kind {
  ORC_OK = 0,
  ORC_CIGAR_TAG_NOT_FOUND = 1,
  ORC_CIGAR_OP_INVALID = 2,
  ORC_PARSE_INT = 3,
  ORC_INVALID_BASE = 4,
  ORC_NOM = 5,
  ORC_PANIC = 6
};

typedef struct {
  int kind;
  char arg[64]; /* the {0} of the error message (op token, digits, base, first 10 bytes) */
} orc_err;

/* struct Cigar sans text, cigar.rs:16-29 (same field order as wga_cigar_counts) */
typedef struct {
  uint64_t match, mismatch, ins_ev, ins_bp, del_ev, del_bp, inv_ins_ev, inv_ins_bp, inv_del_ev,
      inv_del_bp, inv_ev;
} orc_counts;

/* RecStat, common.rs:99-113 */
typedef struct {
  uint64_t aligned_size, matched, mismatched, ins_event, del_event, ins_size, del_size,
      inv_ins_event, inv_ins_size, inv_del_event, inv_del_size, inv_event;
  float inv_size;
} orc_recstat;

/* Formats the reference's error message (errors.rs) into buf. */
void orc_err_message(const orc_err* e, char* buf, size_t cap);

/* cigar.rs:629-707 parse_paf_to_cigar; `cg` includes the "cg:Z:" tag. */
int orc_parse_paf_to_cigar(const char* cg, size_t n, int strand_neg, orc_counts* out, orc_err* err);
/* common.rs:116-140 */
void orc_recstat_from(const orc_counts* c, orc_recstat* out);
/* utils.rs:83-101; out has n bytes */
int orc_reverse_complement(const char* in, size_t n, char* out, orc_err* err);
/* cigar.rs:522-551 parse_cigar_to_insert (+ cigar_unit_insert_seq :492-519).  *t / *q are
 * malloc'd NUL-less buffers that are re-allocated as gaps are inserted. */
int orc_parse_cigar_to_insert(const char* cg, size_t n, char** t, size_t* tn, char** q, size_t* qn,
                              orc_err* err);
/* cigar.rs:344-432 parse_maf_seq_to_cigar (with_h=false).  cigar_text (optional) receives a
 * malloc'd NUL-terminated "<len><op>..." string. */
void orc_parse_maf_seq_to_cigar(const char* t, size_t tn, const char* q, size_t qn, int strand_neg,
                                orc_counts* out, char** cigar_text);
/* cigar.rs:710-741 update_cov_vec */
int orc_update_cov_vec(uint64_t* cov, size_t cov_len, const char* cg, size_t n, size_t start,
                       orc_err* err);
/* cigar.rs:744-804 gen_pesudo_maf_by_cigar; *q is malloc'd and edited in place */
int orc_gen_pesudo_maf_by_cigar(const char* cg, size_t n, char** q, size_t* qn, int base,
                                orc_err* err);
/* paf.rs:159-218 cs_to_cigar (regex restated as a hand scanner); returns malloc'd string */
char* orc_cs_to_cigar(const char* cs, size_t n);

/* caller.rs:159-219; returns safe_end */
size_t orc_find_safe_chunk_boundary(const char* t, const char* q, size_t total, size_t start,
                                    size_t chunk_size, uint64_t svlen_cutoff);
/* caller.rs:388-608 call_within_var on one chunk (rows already cut, starts already advanced).
 * Appends VCF body lines (noodles-vcf 0.43 layout as in README.md:332-342) to *out (malloc'd,
 * NUL-terminated, grown as needed). */
int orc_call_within_var(const char* chro, const char* q_chro, const char* t, const char* q,
                        size_t cols, uint64_t t_start, uint64_t t_end, uint64_t q_start,
                        uint64_t q_end, int strand_neg, int if_snp, uint64_t svlen_cutoff,
                        int if_inv, char** out, size_t* out_len);
/* caller.rs:42-157 per-record driver: chunk loop + create_chunk_record (:221-265) + call. */
int orc_call_var_maf_record(const char* chro, const char* q_chro, const char* t, const char* q,
                            size_t cols, uint64_t t_start, uint64_t t_size_unused,
                            uint64_t q_sline_start, uint64_t q_sline_align, uint64_t q_size,
                            int strand_neg, int if_snp, int if_inv, uint64_t svlen_cutoff,
                            size_t chunk_size, char** out, size_t* out_len);

/* caller.rs:610-822 call_within_var_paf on one PAF record; t_seq / q_seq as fetched by
 * paf.rs:221-237 (inclusive end, no reverse complement); appends VCF body lines to *out. */
int orc_call_within_var_paf(const char* chro, const char* q_chro, const char* cg, size_t cg_n,
                            const char* t_seq, size_t t_n, const char* q_seq, size_t q_n,
                            uint64_t t_start, uint64_t t_end, uint64_t q_start, uint64_t q_end,
                            int strand_neg, int if_snp, uint64_t svlen_cutoff, char** out,
                            size_t* out_len, orc_err* err);

/* cigar.rs:202-245 parse_cigar_to_trim: out = head_ins, head_del, tail_ins, tail_del */
int orc_parse_cigar_to_trim(const char* cg, size_t n, uint64_t out[4], orc_err* err);
/* one record of converter.rs:148-173 paf2chain (header + data lines + "\n\n"); *out malloc'd */
int orc_paf2chain_record(const char* q_name, uint64_t q_size, uint64_t q_start, uint64_t q_end,
                         int strand_neg, const char* t_name, uint64_t t_size, uint64_t t_start,
                         uint64_t t_end, const char* cg, size_t n, uint64_t chain_id, char** out,
                         size_t* out_len, orc_err* err);

/* cigar.rs:155-199 parse_maf_seq_to_trim over the two rows (zip truncates to the shorter) */
void orc_parse_maf_seq_to_trim(const char* t, size_t tn, const char* q, size_t qn, uint64_t out[4]);
/* one record of converter.rs:57-91 maf2chain (chain.rs:103-140 header, cigar.rs:435-457 data lines) */
void orc_maf2chain_record(const char* t_name, uint64_t t_size, uint64_t t_start, uint64_t t_align,
                          const char* q_name, uint64_t q_size, uint64_t q_sline_start,
                          uint64_t q_sline_align, int strand_neg, const char* t, size_t tn,
                          const char* q, size_t qn, uint64_t chain_id, char** out, size_t* out_len);
/* cigar.rs:554-627 parse_chain_to_cigar; lines = n_lines x (size, query_diff, target_diff) */
void orc_parse_chain_to_cigar(const uint64_t* lines, size_t n_lines, int strand_neg, orc_counts* out,
                              char** text);
/* converter.rs:360-388 parse_chain_to_insert; ORC_PANIC where String::insert_str would panic */
int orc_parse_chain_to_insert(const uint64_t* lines, size_t n_lines, char** t, size_t* tn, char** q,
                              size_t* qn);

/* cigar.rs:917-952 parse_cigar_to_base_plotdata (+ emit_baseplotdatas :815-914): *segs = malloc'd
 * n_segs x 5 u64 (ref_start, ref_end, query_start, query_end, kind 0 M / 1 I / 2 D) */
int orc_cigar_to_base_plotdata(const char* cg, size_t n, uint64_t t_start, uint64_t q_start, int strand_neg,
                               uint64_t cutoff, uint64_t** segs, size_t* n_segs, orc_err* err);
/* cigar.rs:955-985 parse_maf_to_base_plotdata */
void orc_maf_to_base_plotdata(const char* t, size_t tn, const char* q, size_t qn, uint64_t t_start,
                              uint64_t q_start, int strand_neg, uint64_t cutoff, uint64_t** segs,
                              size_t* n_segs);

/* cigar.rs:43-75: the (length, op) tokens every consumer folds over, up to the first error; the error and the byte
 * span [*err_off, +*err_len) of the token its message quotes.  `text` = the CIGAR behind "cg:Z:".  Returns the kind. */
int orc_tokenise(const char* text, size_t n, uint64_t* lens, unsigned char* op_first, size_t cap, size_t* n_tok,
                 orc_err* err, size_t* err_off, size_t* err_len);

void orc_free(void* p);
/* test helper: packed ops -> "cg:Z:..." text; returns length (0 if cap too small) */
size_t orc_ops_to_text(const uint32_t* ops, size_t n, char* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
