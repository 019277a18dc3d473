/*
 * oracle.c — CPU restatement of wgatools' CIGAR hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h for the usage rule and the parity-pinning statement: the README VCF golden pins the
 * `call` walk, the data rows of the reference's test/test.html pin the dotplot segment fold; stat /
 * paf2maf / maf2paf / pafcov / pafpseudo / the chain converters are "parity unpinned").
 *
 * Deliberately naive: it keeps the reference's algorithmic structure — text tokenising per
 * consumer, String::insert_str / drain with tail memmove (quadratic), per-base coverage
 * increments — so that it is an independent check of the scan-based GPU kernels and a faithful
 * "port" CPU baseline.  All citations are /root/reference-relative.
 */
#include "oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                               */
/* ------------------------------------------------------------------------------------------ */
static void set_err(orc_err* e, int kind, const char* arg, size_t n) {
  if (!e) return;
  e->kind = kind;
  if (n > sizeof(e->arg) - 1) n = sizeof(e->arg) - 1;
  if (arg && n) memcpy(e->arg, arg, n);
  e->arg[n] = 0;
}

void orc_free(void* p) { free(p); }

/* test helper (not a reference function): packed u32 ops (len << 4 | code) -> "cg:Z:" text */
size_t orc_ops_to_text(const uint32_t* ops, size_t n, char* out, size_t cap) {
  static const char chars[] = "MIDNSHP=XIDB";
  size_t k = 0;
  if (cap < 6) return 0;
  memcpy(out, "cg:Z:", 5);
  k = 5;
  for (size_t i = 0; i < n; i++) {
    if (k + 12 > cap) return 0;
    uint32_t len = ops[i] >> 4, code = ops[i] & 15u;
    char tmp[12];
    int m = 0;
    do {
      tmp[m++] = (char)('0' + len % 10);
      len /= 10;
    } while (len);
    while (m) out[k++] = tmp[--m];
    out[k++] = chars[code < 12 ? code : 11];
  }
  out[k] = 0;
  return k;
}

/* errors.rs:45-74 message templates */
void orc_err_message(const orc_err* e, char* buf, size_t cap) {
  switch (e->kind) {
    case ORC_OK: snprintf(buf, cap, "ok"); break;
    case ORC_CIGAR_TAG_NOT_FOUND: snprintf(buf, cap, "CIGAR start tag not found"); break;
    case ORC_CIGAR_OP_INVALID: snprintf(buf, cap, "CIGAR OP `%s` invalid", e->arg); break;
    case ORC_PARSE_INT: snprintf(buf, cap, "Parse `%s` Into Integer Error", e->arg); break;
    case ORC_INVALID_BASE: snprintf(buf, cap, "Invalid Base: `%s`", e->arg); break;
    case ORC_NOM: /* errors.rs:45 + nom 7 Display of Error<String>: "error {:?} at: {}" */
      snprintf(buf, cap, "Format error Tag at: %s Parse Error by rust::nom, please check", e->arg);
      break;
    default: snprintf(buf, cap, "panic: %s", e->arg); break;
  }
}

static int is_digit(char c) { return c >= '0' && c <= '9'; }

/* number of bytes of the UTF-8 char starting at s (chars() semantics of cst2cu, cigar.rs:45-53) */
static size_t utf8_len(unsigned char c) {
  if (c < 0x80) return 1;
  if ((c >> 5) == 0x6) return 2;
  if ((c >> 4) == 0xE) return 3;
  if ((c >> 3) == 0x1E) return 4;
  return 1;
}

/* One CigarStrTuple (cigar.rs:38): op token and length token. */
typedef struct {
  const char* len;
  size_t len_n;
  const char* op;
  size_t op_n;
} cst_t;

/* parse_cigar_str_tuple, cigar.rs:59-75: empty input -> Eof error (ends fold_many1);
 * else take_while(digit) then take_till(digit). Returns 1 if a tuple was produced. */
static int parse_cigar_str_tuple(const char** p, const char* end, cst_t* out) {
  if (*p == end) return 0;
  const char* s = *p;
  out->len = s;
  while (s < end && is_digit(*s)) s++;
  out->len_n = (size_t)(s - out->len);
  out->op = s;
  while (s < end && !is_digit(*s)) s++;
  out->op_n = (size_t)(s - out->op);
  *p = s;
  return 1;
}

/* cst2cu, cigar.rs:43-56 (+ parse_str2u64, utils.rs:69-74).  The op is checked before the
 * length. On success *op points at the single (possibly multi-byte) char. */
static int cst2cu(const cst_t* t, const char** op, size_t* op_n, uint64_t* len, orc_err* err) {
  if (t->op_n == 0) { /* chars.next() == None */
    set_err(err, ORC_CIGAR_OP_INVALID, "", 0);
    return ORC_CIGAR_OP_INVALID;
  }
  size_t first = utf8_len((unsigned char)t->op[0]);
  if (first < t->op_n) { /* a second char exists */
    set_err(err, ORC_CIGAR_OP_INVALID, t->op, t->op_n);
    return ORC_CIGAR_OP_INVALID;
  }
  if (t->len_n == 0) { /* "".parse::<u64>() fails */
    set_err(err, ORC_PARSE_INT, "", 0);
    return ORC_PARSE_INT;
  }
  uint64_t v = 0;
  for (size_t i = 0; i < t->len_n; i++) {
    uint64_t d = (uint64_t)(t->len[i] - '0');
    if (v > (UINT64_MAX - d) / 10) { /* u64 overflow -> parse error */
      set_err(err, ORC_PARSE_INT, t->len, t->len_n);
      return ORC_PARSE_INT;
    }
    v = v * 10 + d;
  }
  *op = t->op;
  *op_n = t->op_n;
  *len = v;
  return ORC_OK;
}

/* The token stream every reference consumer folds over (fold_many1(parse_cigar_str_tuple) + cst2cu, cigar.rs:43-75):
 * the (length, op char) pairs up to the first error, the error that ends the fold, and the byte span of the token it
 * quotes.  `text` is the CIGAR behind the "cg:Z:" tag.  op_first[] gets the first byte of each op char. */
int orc_tokenise(const char* text, size_t n, uint64_t* lens, unsigned char* op_first, size_t cap, size_t* n_tok,
                 orc_err* err, size_t* err_off, size_t* err_len) {
  const char* p = text;
  const char* end = text + n;
  size_t k = 0;
  int rc = ORC_OK;
  if (err) err->kind = ORC_OK;
  if (err_off) *err_off = 0;
  if (err_len) *err_len = 0;
  if (n == 0) { /* fold_many1 on no input: Many1 error -> errors.rs:92 slices input[..10] of an empty string: panic */
    if (err) set_err(err, ORC_PANIC, "", 0);
    if (n_tok) *n_tok = 0;
    return ORC_PANIC;
  }
  cst_t t;
  while (parse_cigar_str_tuple(&p, end, &t)) {
    const char* op;
    size_t op_n;
    uint64_t len;
    orc_err e;
    rc = cst2cu(&t, &op, &op_n, &len, &e);
    if (rc) {
      if (err) *err = e;
      if (rc == ORC_CIGAR_OP_INVALID) {
        if (err_off) *err_off = (size_t)(t.op - text);
        if (err_len) *err_len = t.op_n;
      } else { /* ParseIntError quotes the length token (empty, or the overflowing digits) */
        if (err_off) *err_off = (size_t)(t.len - text);
        if (err_len) *err_len = t.len_n;
      }
      break;
    }
    if (k < cap) {
      lens[k] = len;
      op_first[k] = (unsigned char)op[0];
    }
    k++;
  }
  if (n_tok) *n_tok = k;
  return rc;
}

/* tag("cg:Z:") + the From<nom::Err> conversion of errors.rs:88-96 (slices input[..10]). */
static int strip_tag(const char** p, const char* end, orc_err* err) {
  size_t n = (size_t)(end - *p);
  if (n >= 5 && memcmp(*p, "cg:Z:", 5) == 0) {
    *p += 5;
    return ORC_OK;
  }
  if (n < 10) {
    set_err(err, ORC_PANIC, "byte index 10 is out of bounds (errors.rs:92)", 46);
    return ORC_PANIC;
  }
  set_err(err, ORC_NOM, *p, 10);
  return ORC_NOM;
}

/* fold_many1 needs at least one tuple; on an empty CIGAR the Many1 error is converted through
 * errors.rs:92 with an empty input -> the reference panics. */
static int empty_cigar_panic(orc_err* err) {
  set_err(err, ORC_PANIC, "byte index 10 is out of bounds (errors.rs:92)", 46);
  return ORC_PANIC;
}

static int op_is(const char* op, size_t n, char c) { return n == 1 && op[0] == c; }

/* ------------------------------------------------------------------------------------------ */
/* parse_paf_to_cigar, cigar.rs:629-707                                                        */
/* ------------------------------------------------------------------------------------------ */
int orc_parse_paf_to_cigar(const char* cg, size_t n, int strand_neg, orc_counts* out,
                           orc_err* err) {
  memset(out, 0, sizeof(*out));
  if (err) err->kind = ORC_OK;
  const char* p = cg;
  const char* end = cg + n;
  int inv = 0;
  if (strand_neg) { /* :643-649 */
    out->inv_ev = 1;
    inv = 1;
  }
  int rc = strip_tag(&p, end, err); /* :652 */
  if (rc) return rc;
  if (p == end) return empty_cigar_panic(err);
  cst_t t;
  while (parse_cigar_str_tuple(&p, end, &t)) { /* :654-691 */
    const char* op;
    size_t op_n;
    uint64_t len;
    rc = cst2cu(&t, &op, &op_n, &len, err);
    if (rc) return rc; /* first error sticks; later ops are skipped and `res?` returns it */
    if (op_is(op, op_n, 'M') || op_is(op, op_n, '=')) {
      out->match += len;
    } else if (op_is(op, op_n, 'X')) {
      out->mismatch += len;
    } else if (op_is(op, op_n, 'I')) {
      if (inv) {
        out->inv_ins_ev += 1;
        out->inv_ins_bp += len;
      } else {
        out->ins_ev += 1;
        out->ins_bp += len;
      }
    } else if (op_is(op, op_n, 'D')) {
      if (inv) {
        out->inv_del_ev += 1;
        out->inv_del_bp += len;
      } else {
        out->del_ev += 1;
        out->del_bp += len;
      }
    } else {
      set_err(err, ORC_CIGAR_OP_INVALID, op, op_n);
      return ORC_CIGAR_OP_INVALID;
    }
  }
  return ORC_OK;
}

/* RecStat::from, common.rs:116-140 */
void orc_recstat_from(const orc_counts* c, orc_recstat* r) {
  memset(r, 0, sizeof(*r));
  r->matched = c->match;
  r->mismatched = c->mismatch;
  r->ins_event = c->ins_ev;
  r->del_event = c->del_ev;
  r->ins_size = c->ins_bp;
  r->del_size = c->del_bp;
  r->inv_ins_event = c->inv_ins_ev;
  r->inv_ins_size = c->inv_ins_bp;
  r->inv_del_event = c->inv_del_ev;
  r->inv_del_size = c->inv_del_bp;
  r->aligned_size = r->matched + r->mismatched + r->del_size + r->inv_del_size;
  uint64_t query_align_size = r->matched + r->mismatched + r->ins_size + r->inv_ins_size;
  r->inv_event = c->inv_ev;
  if (r->inv_event != 0) { /* usize as f32 / usize as f32 */
    r->inv_size = (float)(r->aligned_size + query_align_size) / (float)(r->inv_event + 1);
  }
}

/* reverse_complement, utils.rs:83-101: iterates chars().rev(); first offender (from the END of
 * the input) is reported. */
int orc_reverse_complement(const char* in, size_t n, char* out, orc_err* err) {
  if (err) err->kind = ORC_OK;
  for (size_t i = 0; i < n; i++) {
    char c = in[n - 1 - i];
    char o;
    switch (c) {
      case 'A': o = 'T'; break;
      case 'C': o = 'G'; break;
      case 'G': o = 'C'; break;
      case 'T': o = 'A'; break;
      case 'N': o = 'N'; break;
      case 'a': o = 't'; break;
      case 'c': o = 'g'; break;
      case 'g': o = 'c'; break;
      case 't': o = 'a'; break;
      case 'n': o = 'n'; break;
      default:
        set_err(err, ORC_INVALID_BASE, &c, 1);
        return ORC_INVALID_BASE;
    }
    out[i] = o;
  }
  return ORC_OK;
}

/* String::insert_str(idx, "-" * count): panics when idx > len; tail memmove otherwise. */
static int string_insert_gaps(char** s, size_t* sn, uint64_t idx, uint64_t count, orc_err* err) {
  if (idx > *sn) {
    set_err(err, ORC_PANIC, "String::insert_str index out of range (cigar.rs:507,513)", 56);
    return ORC_PANIC;
  }
  char* ns = (char*)realloc(*s, *sn + count + 1);
  if (!ns) abort();
  memmove(ns + idx + count, ns + idx, *sn - idx);
  memset(ns + idx, '-', count);
  *s = ns;
  *sn += count;
  return ORC_OK;
}

/* parse_cigar_to_insert, cigar.rs:522-551, with cigar_unit_insert_seq, :492-519 */
int orc_parse_cigar_to_insert(const char* cg, size_t n, char** t, size_t* tn, char** q, size_t* qn,
                              orc_err* err) {
  if (err) err->kind = ORC_OK;
  const char* p = cg;
  const char* end = cg + n;
  int rc = strip_tag(&p, end, err); /* :529 */
  if (rc) return rc;
  if (p == end) return empty_cigar_panic(err);
  uint64_t current_offset = 0;
  cst_t tok;
  while (parse_cigar_str_tuple(&p, end, &tok)) {
    const char* op;
    size_t op_n;
    uint64_t len;
    rc = cst2cu(&tok, &op, &op_n, &len, err);
    if (rc) return rc;
    if (op_is(op, op_n, 'M') || op_is(op, op_n, '=') || op_is(op, op_n, 'X')) {
      current_offset += len; /* :500-503 */
    } else if (op_is(op, op_n, 'I')) { /* :504-509 insert '-' into target */
      rc = string_insert_gaps(t, tn, current_offset, len, err);
      if (rc) return rc;
      current_offset += len;
    } else if (op_is(op, op_n, 'D')) { /* :510-515 insert '-' into query */
      rc = string_insert_gaps(q, qn, current_offset, len, err);
      if (rc) return rc;
      current_offset += len;
    } else {
      set_err(err, ORC_CIGAR_OP_INVALID, op, op_n);
      return ORC_CIGAR_OP_INVALID;
    }
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* MAF column pair walk                                                                        */
/* ------------------------------------------------------------------------------------------ */
/* cigar_cat_ext, cigar.rs:298-308 */
static char cigar_cat_ext(char c1, char c2) {
  if (c1 == c2) return '=';
  if (c1 == '-') return 'I';
  if (c2 == '-') return 'D';
  return 'X';
}
/* cigar_cat_ext_caller, cigar.rs:314-328 */
static char cigar_cat_ext_caller(char c1, char c2) {
  if (c1 == '-') return c2 == '-' ? 'W' : 'I';
  if (c2 == '-') return 'D';
  return c1 == c2 ? '=' : 'X';
}

typedef struct {
  char* s;
  size_t n, cap;
} sbuf;
static void sb_push(sbuf* b, const char* s, size_t n) {
  if (b->n + n + 1 > b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 256;
    while (nc < b->n + n + 1) nc *= 2;
    b->s = (char*)realloc(b->s, nc);
    if (!b->s) abort();
    b->cap = nc;
  }
  memcpy(b->s + b->n, s, n);
  b->n += n;
  b->s[b->n] = 0;
}
static void sb_printf_u64(sbuf* b, uint64_t v) {
  char tmp[32];
  int k = snprintf(tmp, sizeof tmp, "%llu", (unsigned long long)v);
  sb_push(b, tmp, (size_t)k);
}

/* parse_maf_seq_to_cigar(rec, with_h = false), cigar.rs:344-432.  The rows are zipped (shorter
 * wins) and grouped into maximal runs of equal category (itertools group_by). */
void orc_parse_maf_seq_to_cigar(const char* t, size_t tn, const char* q, size_t qn, int strand_neg,
                                orc_counts* out, char** cigar_text) {
  memset(out, 0, sizeof(*out));
  sbuf sb = {0, 0, 0};
  sb_push(&sb, "", 0);
  size_t cols = tn < qn ? tn : qn;
  int inv = 0;
  if (strand_neg) { /* :370-376 */
    out->inv_ev = 1;
    inv = 1;
  }
  size_t i = 0;
  while (i < cols) {
    char k = cigar_cat_ext(t[i], q[i]);
    size_t j = i + 1;
    while (j < cols && cigar_cat_ext(t[j], q[j]) == k) j++;
    uint64_t len = (uint64_t)(j - i);
    switch (k) { /* :382-408 */
      case '=': out->match += len; break;
      case 'I':
        if (inv) {
          out->inv_ins_ev += 1;
          out->inv_ins_bp += len;
        } else {
          out->ins_ev += 1;
          out->ins_bp += len;
        }
        break;
      case 'D':
        if (inv) {
          out->inv_del_ev += 1;
          out->inv_del_bp += len;
        } else {
          out->del_ev += 1;
          out->del_bp += len;
        }
        break;
      case 'X': out->mismatch += len; break;
    }
    sb_printf_u64(&sb, len); /* :409-410 */
    sb_push(&sb, &k, 1);
    i = j;
  }
  if (cigar_text)
    *cigar_text = sb.s;
  else
    free(sb.s);
}

/* ------------------------------------------------------------------------------------------ */
/* update_cov_vec, cigar.rs:710-741                                                            */
/* ------------------------------------------------------------------------------------------ */
int orc_update_cov_vec(uint64_t* cov, size_t cov_len, const char* cg, size_t n, size_t start,
                       orc_err* err) {
  if (err) err->kind = ORC_OK;
  const char* p = cg;
  const char* end = cg + n;
  int rc = strip_tag(&p, end, err);
  if (rc) return rc;
  if (p == end) return empty_cigar_panic(err);
  size_t pos = start;
  cst_t tok;
  while (parse_cigar_str_tuple(&p, end, &tok)) {
    const char* op;
    size_t op_n;
    uint64_t len;
    rc = cst2cu(&tok, &op, &op_n, &len, err);
    if (rc) return rc;
    size_t length = (size_t)len;
    if (op_is(op, op_n, 'M') || op_is(op, op_n, '=')) { /* :721-728 per-base increments */
      for (size_t i = pos; i < pos + length; i++) {
        if (i < cov_len) cov[i] += 1;
      }
      pos += length;
    } else if (op_is(op, op_n, 'I') || op_is(op, op_n, 'S')) {
      /* :729 nothing */
    } else {
      pos += length; /* :731-733 everything else (D, X, N, H, P, ...) just moves */
    }
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* gen_pesudo_maf_by_cigar, cigar.rs:744-804                                                   */
/* ------------------------------------------------------------------------------------------ */
static void str_push_n(char** s, size_t* sn, char c, size_t count) {
  char* ns = (char*)realloc(*s, *sn + count + 1);
  if (!ns) abort();
  memset(ns + *sn, c, count);
  *s = ns;
  *sn += count;
}

int orc_gen_pesudo_maf_by_cigar(const char* cg, size_t n, char** q, size_t* qn, int base,
                                orc_err* err) {
  if (err) err->kind = ORC_OK;
  const char* p = cg;
  const char* end = cg + n;
  int rc = strip_tag(&p, end, err);
  if (rc) return rc;
  if (p == end) return empty_cigar_panic(err);
  size_t current_offset = 0;
  cst_t tok;
  while (parse_cigar_str_tuple(&p, end, &tok)) {
    const char* op;
    size_t op_n;
    uint64_t len;
    rc = cst2cu(&tok, &op, &op_n, &len, err);
    if (rc) return rc;
    size_t length = (size_t)len;
    if (op_is(op, op_n, 'M') || op_is(op, op_n, '=')) { /* :760-768 */
      if (base)
        current_offset += length;
      else
        str_push_n(q, qn, '1', length);
    } else if (op_is(op, op_n, 'I') || op_is(op, op_n, 'S')) { /* :769-776 drain */
      if (base) {
        if (current_offset + length > *qn || current_offset > current_offset + length) {
          set_err(err, ORC_PANIC, "String::drain range out of bounds (cigar.rs:772)", 48);
          return ORC_PANIC;
        }
        memmove(*q + current_offset, *q + current_offset + length,
                *qn - current_offset - length);
        *qn -= length;
      }
    } else if (op_is(op, op_n, 'D')) { /* :777-786 */
      if (base) {
        rc = string_insert_gaps(q, qn, current_offset, length, err);
        if (rc) return rc;
        current_offset += length;
      } else {
        str_push_n(q, qn, '-', length);
      }
    } else if (op_is(op, op_n, 'X')) { /* :787-795 */
      if (base)
        current_offset += length;
      else
        str_push_n(q, qn, '0', length);
    } else {
      /* :796 ignored */
    }
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* cs_to_cigar, paf.rs:159-218.  Regex (:[0-9]+|\*[a-z][a-z]|[=\+\-][A-Za-z]+) restated as a   */
/* leftmost scanner: unmatched bytes are skipped exactly like captures_iter does.              */
/* ------------------------------------------------------------------------------------------ */
static int is_lower(char c) { return c >= 'a' && c <= 'z'; }
static int is_alpha(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

char* orc_cs_to_cigar(const char* cs, size_t n) {
  sbuf sb = {0, 0, 0};
  sb_push(&sb, "", 0);
  char last_op = 'M';
  uint64_t last_len = 0;
  size_t i = 0;
  while (i < n) {
    char c = cs[i];
    if (c == ':' && i + 1 < n && is_digit(cs[i + 1])) {
      size_t j = i + 1;
      uint64_t length = 0;
      while (j < n && is_digit(cs[j])) length = length * 10 + (uint64_t)(cs[j++] - '0');
      if (last_op == 'M') {
        last_len += length;
      } else {
        if (last_len > 0) {
          sb_printf_u64(&sb, last_len);
          sb_push(&sb, &last_op, 1);
        }
        last_op = 'M';
        last_len = length;
      }
      i = j;
    } else if (c == '*' && i + 2 < n + 0 && is_lower(cs[i + 1]) && is_lower(cs[i + 2])) {
      if (last_op == 'X') {
        last_len += 1;
      } else {
        if (last_len > 0) {
          sb_printf_u64(&sb, last_len);
          sb_push(&sb, &last_op, 1);
        }
        last_op = 'X';
        last_len = 1;
      }
      i += 3;
    } else if ((c == '=' || c == '+' || c == '-') && i + 1 < n && is_alpha(cs[i + 1])) {
      size_t j = i + 1;
      while (j < n && is_alpha(cs[j])) j++;
      uint64_t length = (uint64_t)(j - i - 1);
      if (c == '-' || c == '+') {
        if (last_len > 0) {
          sb_printf_u64(&sb, last_len);
          sb_push(&sb, &last_op, 1);
        }
        sb_printf_u64(&sb, length);
        sb_push(&sb, c == '-' ? "D" : "I", 1);
        last_len = 0;
        last_op = 'M';
      } /* '=' parts are matched but ignored, paf.rs:209 */
      i = j;
    } else {
      i++;
    }
  }
  if (last_len > 0) {
    sb_printf_u64(&sb, last_len);
    sb_push(&sb, &last_op, 1);
  }
  return sb.s;
}

/* ------------------------------------------------------------------------------------------ */
/* call (MAF), caller.rs                                                                       */
/* ------------------------------------------------------------------------------------------ */
/* find_safe_chunk_boundary, caller.rs:159-219 */
size_t orc_find_safe_chunk_boundary(const char* t, const char* q, size_t total, size_t start,
                                    size_t chunk_size, uint64_t svlen_cutoff) {
  size_t proposed_end = start + chunk_size < total ? start + chunk_size : total;
  size_t current_gap_size = 0;
  int in_sv = 0;
  size_t sv_start = 0;
  size_t safe_end = proposed_end;
  for (size_t abs_pos = start; abs_pos < proposed_end; abs_pos++) {
    char rc = t[abs_pos], qc = q[abs_pos];
    if (rc == '-' || qc == '-') {
      if (!in_sv) {
        in_sv = 1;
        sv_start = abs_pos;
      }
      current_gap_size += 1;
    } else if (in_sv) {
      if (current_gap_size >= (size_t)svlen_cutoff) {
        if (sv_start >= start) safe_end = abs_pos;
      }
      in_sv = 0;
      current_gap_size = 0;
    }
  }
  if (in_sv && current_gap_size >= (size_t)svlen_cutoff) { /* :202-215 */
    size_t end_pos = proposed_end;
    for (size_t pos = proposed_end; pos < total; pos++) {
      if (t[pos] != '-' && q[pos] != '-') {
        end_pos = pos;
        break;
      }
    }
    safe_end = end_pos;
  }
  return safe_end;
}

/* noodles-vcf 0.43 reference/alternate base parsing upper-cases a/c/g/t/n (third-party, source
 * not in the reference tree: unpinned). */
static void sb_push_bases_upper(sbuf* b, const char* s, size_t n) {
  for (size_t i = 0; i < n; i++) {
    char c = s[i];
    if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    sb_push(b, &c, 1);
  }
}

/* one VCF body line in noodles-vcf 0.43 layout (README.md:332-342):
 * CHROM POS ID(.) REF ALT QUAL(.) FILTER(.) INFO FORMAT SAMPLE */
static void emit_vcf(sbuf* out, const char* chro, uint64_t pos, const char* ref, size_t ref_n,
                     const char* alt, size_t alt_n, int alt_symbolic, const char* info,
                     const char* fmt_and_sample) {
  sb_push(out, chro, strlen(chro));
  sb_push(out, "\t", 1);
  sb_printf_u64(out, pos);
  sb_push(out, "\t.\t", 3);
  sb_push_bases_upper(out, ref, ref_n);
  sb_push(out, "\t", 1);
  if (alt_symbolic)
    sb_push(out, alt, alt_n);
  else
    sb_push_bases_upper(out, alt, alt_n);
  sb_push(out, "\t.\t.\t", 5);
  if (info)
    sb_push(out, info, strlen(info));
  else
    sb_push(out, ".", 1);
  sb_push(out, "\t", 1);
  sb_push(out, fmt_and_sample, strlen(fmt_and_sample));
  sb_push(out, "\n", 1);
}

/* call_within_var, caller.rs:388-608 */
int orc_call_within_var(const char* chro, const char* q_chro, const char* t, const char* q,
                        size_t cols, uint64_t t_start, uint64_t t_end, uint64_t q_start,
                        uint64_t q_end, int strand_neg, int if_snp, uint64_t svlen_cutoff,
                        int if_inv, char** out, size_t* out_len) {
  sbuf sb = {*out, *out_len, *out ? *out_len + 1 : 0};
  if (!sb.s) sb_push(&sb, "", 0);
  /* :411-415 gap-stripped copies */
  char* t_ref = (char*)malloc(cols + 1);
  char* q_ref = (char*)malloc(cols + 1);
  size_t t_ref_n = 0, q_ref_n = 0;
  for (size_t i = 0; i < cols; i++)
    if (t[i] != '-') t_ref[t_ref_n++] = t[i];
  for (size_t i = 0; i < cols; i++)
    if (q[i] != '-') q_ref[q_ref_n++] = q[i];
  uint64_t target_current_offset = t_start;
  uint64_t query_current_offset = q_start;
  char suffix = strand_neg ? 'N' : 'P';
  char info[256], qi[512];
  int rc = 0;
  if (strand_neg && t_ref_n != 0 && if_inv) { /* :423-440 */
    snprintf(info, sizeof info, "SVTYPE=INV;END=%llu", (unsigned long long)t_end);
    snprintf(qi, sizeof qi, "GT:QI\t1|1:%s@%llu@%llu@%c", q_chro, (unsigned long long)q_start,
             (unsigned long long)q_end, suffix);
    emit_vcf(&sb, chro, target_current_offset + 1, t_ref, 1, "<INV>", 5, 1, info, qi);
  }
  const char* init_info = strand_neg ? "INV_NEST=TRUE;" : ""; /* :448-451 */
  int after_m = 0;
  size_t i = 0;
  while (i < cols) { /* group_by(cigar_cat_ext_caller) :444-446 */
    char k = cigar_cat_ext_caller(t[i], q[i]);
    size_t j = i + 1;
    while (j < cols && cigar_cat_ext_caller(t[j], q[j]) == k) j++;
    uint64_t len = (uint64_t)(j - i);
    i = j;
    switch (k) {
      case '=': /* :456-460 */
        target_current_offset += len;
        query_current_offset += len;
        after_m = 1;
        break;
      case 'W': break; /* :461-463 */
      case 'I':        /* :464-515 */
        if (len > svlen_cutoff) {
          if (!after_m) {
            query_current_offset += len;
            after_m = 0;
            continue;
          }
          uint64_t ts = target_current_offset - t_start - 1;
          uint64_t qs = query_current_offset - q_start - 1;
          if (ts + 1 > t_ref_n || qs + len + 1 > q_ref_n) {
            rc = ORC_PANIC;
            goto done;
          }
          snprintf(info, sizeof info, "%sSVTYPE=INS;SVLEN=%llu;END=%llu", init_info,
                   (unsigned long long)len, (unsigned long long)target_current_offset);
          snprintf(qi, sizeof qi, "GT:QI\t1|1:%s@%llu@%llu@%c", q_chro,
                   (unsigned long long)query_current_offset,
                   (unsigned long long)(query_current_offset + len), suffix);
          emit_vcf(&sb, chro, target_current_offset, t_ref + ts, 1, q_ref + qs, (size_t)len + 1, 0,
                   info, qi);
        }
        query_current_offset += len;
        after_m = 0;
        break;
      case 'D': /* :516-569 */
        if (len > svlen_cutoff) {
          if (!after_m) {
            target_current_offset += len;
            after_m = 0;
            continue;
          }
          uint64_t ts = target_current_offset - t_start - 1;
          uint64_t qs = query_current_offset - q_start - 1;
          if (ts + len + 1 > t_ref_n || qs + 1 > q_ref_n) {
            rc = ORC_PANIC;
            goto done;
          }
          snprintf(info, sizeof info, "%sSVTYPE=DEL;SVLEN=%llu;END=%llu", init_info,
                   (unsigned long long)len, (unsigned long long)(target_current_offset + len));
          snprintf(qi, sizeof qi, "GT:QI\t1|1:%s@%llu@%llu@%c", q_chro,
                   (unsigned long long)query_current_offset,
                   (unsigned long long)query_current_offset, suffix);
          emit_vcf(&sb, chro, target_current_offset, t_ref + ts, (size_t)len + 1, q_ref + qs, 1, 0,
                   info, qi);
        }
        target_current_offset += len;
        after_m = 0;
        break;
      case 'X': /* :570-603 */
        if (if_snp) {
          for (uint64_t s = 0; s < len; s++) {
            uint64_t ts = target_current_offset - t_start;
            uint64_t qs = query_current_offset - q_start;
            if (ts + 1 > t_ref_n || qs + 1 > q_ref_n) {
              rc = ORC_PANIC;
              goto done;
            }
            snprintf(qi, sizeof qi, "GT:QI\t1|1:%s@%llu@%c", q_chro,
                     (unsigned long long)query_current_offset, suffix);
            emit_vcf(&sb, chro, target_current_offset + 1, t_ref + ts, 1, q_ref + qs, 1, 0, NULL,
                     qi);
            target_current_offset += 1;
            query_current_offset += 1;
          }
        } else {
          query_current_offset += len;
          target_current_offset += len;
        }
        after_m = 1;
        break;
    }
  }
done:
  free(t_ref);
  free(q_ref);
  *out = sb.s;
  *out_len = sb.n;
  return rc;
}

/* call_within_var_paf, caller.rs:610-822.  t_seq / q_seq are what target_seq_with_fa /
 * query_seq_with_fa (paf.rs:221-237) fetched: [start, end] INCLUSIVE, forward strand, no
 * reverse complement for '-'.  `cg` includes the "cg:Z:" tag.  Errors raised inside the fold are
 * kept in its accumulator (which stops the walk) and then discarded (:673, :815-819); slices
 * out of range panic. */
int orc_call_within_var_paf(const char* chro, const char* q_chro, const char* cg, size_t cg_n,
                            const char* t_seq, size_t t_n, const char* q_seq, size_t q_n,
                            uint64_t t_start, uint64_t t_end, uint64_t q_start, uint64_t q_end,
                            int strand_neg, int if_snp, uint64_t svlen_cutoff, char** out,
                            size_t* out_len, orc_err* err) {
  sbuf sb = {*out, *out_len, *out ? *out_len + 1 : 0};
  if (!sb.s) sb_push(&sb, "", 0);
  char suffix = strand_neg ? 'N' : 'P';
  char info[256], qi[512];
  int rc = 0;
  if (strand_neg) { /* :640-658, no -i gate */
    if (t_n < 1) {
      rc = ORC_PANIC;
      goto done;
    }
    snprintf(info, sizeof info, "SVTYPE=INV;END=%llu", (unsigned long long)t_end);
    snprintf(qi, sizeof qi, "GT:QI\t1|1:%s@%llu@%llu@%c", q_chro, (unsigned long long)q_start,
             (unsigned long long)q_end, suffix);
    emit_vcf(&sb, chro, t_start + 1, t_seq, 1, "<INV>", 5, 1, info, qi);
  }
  const char* p = cg;
  const char* end = cg + cg_n;
  rc = strip_tag(&p, end, err); /* :662 */
  if (rc) goto done;
  if (p == end) { /* fold_many1 without a single tuple: Many1 error -> errors.rs:92 */
    rc = ORC_PANIC;
    goto done;
  }
  uint64_t t_pos = t_start, q_pos = q_start;
  const char* init_info = strand_neg ? "INV_NEST=TRUE;" : "";
  int after_m = 0;
  cst_t tok;
  while (parse_cigar_str_tuple(&p, end, &tok)) {
    const char* op;
    size_t op_n;
    uint64_t len;
    orc_err e2;
    if (cst2cu(&tok, &op, &op_n, &len, &e2)) break; /* res = Err: the remaining ops are skipped */
    char c = op_n == 1 ? op[0] : '?';
    if (c == 'M' || c == '=') {
      t_pos += len;
      q_pos += len;
      after_m = 1;
    } else if (c == 'X') {
      if (if_snp) {
        for (uint64_t k = 0; k < len; k++) {
          uint64_t ts = t_pos - t_start, qs = q_pos - q_start;
          if (ts + 1 > t_n || qs + 1 > q_n) {
            rc = ORC_PANIC;
            goto done;
          }
          snprintf(qi, sizeof qi, "GT:QI\t1|1:%s@%llu@%c", q_chro, (unsigned long long)q_pos, suffix);
          emit_vcf(&sb, chro, t_pos + 1, t_seq + ts, 1, q_seq + qs, 1, 0, NULL, qi);
          t_pos++;
          q_pos++;
        }
      } else {
        t_pos += len;
        q_pos += len;
      }
      after_m = 1;
    } else if (c == 'I') {
      if (len > svlen_cutoff) {
        if (!after_m) {
          q_pos += len;
          after_m = 0;
          continue;
        }
        uint64_t ts = t_pos - t_start - 1, qs = q_pos - q_start - 1;
        if (ts + 1 > t_n || qs + len + 1 > q_n) {
          rc = ORC_PANIC;
          goto done;
        }
        snprintf(info, sizeof info, "%sSVTYPE=INS;SVLEN=%llu;END=%llu", init_info,
                 (unsigned long long)len, (unsigned long long)t_pos);
        snprintf(qi, sizeof qi, "GT:QI\t1|1:%s@%llu@%llu@%c", q_chro, (unsigned long long)q_pos,
                 (unsigned long long)(q_pos + len), suffix);
        emit_vcf(&sb, chro, t_pos, t_seq + ts, 1, q_seq + qs, (size_t)len + 1, 0, info, qi);
      }
      q_pos += len;
      after_m = 0;
    } else if (c == 'D') {
      if (len > svlen_cutoff) {
        if (!after_m) {
          t_pos += len;
          after_m = 0;
          continue;
        }
        uint64_t ts = t_pos - t_start - 1, qs = q_pos - q_start - 1;
        if (ts + len + 1 > t_n || qs + 1 > q_n) {
          rc = ORC_PANIC;
          goto done;
        }
        snprintf(info, sizeof info, "%sSVTYPE=DEL;SVLEN=%llu;END=%llu", init_info,
                 (unsigned long long)len, (unsigned long long)(t_pos + len));
        snprintf(qi, sizeof qi, "GT:QI\t1|1:%s@%llu@%llu@%c", q_chro, (unsigned long long)q_pos,
                 (unsigned long long)q_pos, suffix);
        emit_vcf(&sb, chro, t_pos, t_seq + ts, (size_t)len + 1, q_seq + qs, 1, 0, info, qi);
      }
      t_pos += len;
      after_m = 0;
    } else {
      break; /* CigarOpInvalid kept in the accumulator, discarded after the fold */
    }
  }
done:
  if (rc == ORC_PANIC && err) set_err(err, ORC_PANIC, "slice", 5);
  *out = sb.s;
  *out_len = sb.n;
  return rc;
}

/* ---- paf2chain ---------------------------------------------------------
 * parse_cigar_to_trim, cigar.rs:202-245: I / D bases before the first M-like op; tail = length of
 * the LAST I / D op behind the last M-like op (assignment, not accumulation). */
int orc_parse_cigar_to_trim(const char* cg, size_t n, uint64_t out[4], orc_err* err) {
  uint64_t head_ins = 0, head_del = 0, tail_ins = 0, tail_del = 0;
  int head_indel = 1;
  const char* p = cg;
  const char* end = cg + n;
  int rc = strip_tag(&p, end, err);
  if (rc) return rc;
  if (p == end) {
    set_err(err, ORC_PANIC, "empty", 5);
    return ORC_PANIC;
  }
  cst_t tok;
  while (parse_cigar_str_tuple(&p, end, &tok)) {
    const char* op;
    size_t op_n;
    uint64_t len;
    rc = cst2cu(&tok, &op, &op_n, &len, err);
    if (rc) return rc;
    char c = op_n == 1 ? op[0] : '?';
    if (c == 'M' || c == '=' || c == 'X') {
      tail_ins = 0;
      tail_del = 0;
      head_indel = 0;
    } else if (c == 'I') {
      if (head_indel) head_ins += len;
      tail_ins = len;
    } else if (c == 'D') {
      if (head_indel) head_del += len;
      tail_del = len;
    } else {
      set_err(err, ORC_CIGAR_OP_INVALID, op, op_n);
      return ORC_CIGAR_OP_INVALID;
    }
  }
  out[0] = head_ins;
  out[1] = head_del;
  out[2] = tail_ins;
  out[3] = tail_del;
  return ORC_OK;
}

/* one record of converter::paf2chain (converter.rs:148-173): ChainHeader::try_from(&PafRecord)
 * (chain.rs:142-183, incl. the '-' strand arithmetic that reuses the already updated start), its
 * Display (:185-203, score 255 prints as "255"), parse_cigar_to_chain + cigar_unit_chain
 * (cigar.rs:251-295,460-490) and the closing "\n\n".  *out is malloc'd. */
int orc_paf2chain_record(const char* q_name, uint64_t q_size, uint64_t q_start, uint64_t q_end,
                         int strand_neg, const char* t_name, uint64_t t_size, uint64_t t_start,
                         uint64_t t_end, const char* cg, size_t n, uint64_t chain_id, char** out,
                         size_t* out_len, orc_err* err) {
  uint64_t tr[4];
  int rc = orc_parse_cigar_to_trim(cg, n, tr, err);
  if (rc) return rc;
  uint64_t qs = q_start, qe = q_end, ts = t_start, te = t_end;
  if (!strand_neg) {
    qs += tr[0];
    ts += tr[1];
    qe -= tr[2];
    te -= tr[3];
  } else {
    ts += tr[1];
    te -= tr[3];
    qs = q_size - (qe - tr[0]);
    qe = q_size - (qs + tr[2]);
  }
  sbuf sb = {NULL, 0, 0};
  char line[1024];
  snprintf(line, sizeof line, "chain\t255\t%s\t%llu\t+\t%llu\t%llu\t%s\t%llu\t%c\t%llu\t%llu\t%llu", t_name,
           (unsigned long long)t_size, (unsigned long long)ts, (unsigned long long)te, q_name,
           (unsigned long long)q_size, strand_neg ? '-' : '+', (unsigned long long)qs,
           (unsigned long long)qe, (unsigned long long)chain_id);
  sb_push(&sb, line, strlen(line));
  const char* p = cg;
  const char* end = cg + n;
  strip_tag(&p, end, err);
  uint64_t size = 0, qd = 0, td = 0;
  cst_t tok;
  while (parse_cigar_str_tuple(&p, end, &tok)) {
    const char* op;
    size_t op_n;
    uint64_t len;
    if (cst2cu(&tok, &op, &op_n, &len, err)) break; /* cannot happen: the trim pass accepted the text */
    char c = op[0];
    if (c == 'M' || c == 'X' || c == '=') {
      if (size != 0 && td + qd != 0) {
        snprintf(line, sizeof line, "\n%llu\t%llu\t%llu", (unsigned long long)size,
                 (unsigned long long)qd, (unsigned long long)td);
        sb_push(&sb, line, strlen(line));
        size = 0;
      }
      size += len;
      td = 0;
      qd = 0;
    } else if (c == 'I') {
      td += len;
    } else {
      qd += len;
    }
  }
  snprintf(line, sizeof line, "\n%llu\n\n", (unsigned long long)size);
  sb_push(&sb, line, strlen(line));
  *out = sb.s;
  *out_len = sb.n;
  return ORC_OK;
}

/* ---- the other chain converters ------------------------------------------- */
/* cigar_cat, cigar.rs:331-341: equal or both non-gap -> 'M' */
static char cigar_cat(char c1, char c2) {
  if (c1 == c2) return 'M';
  if (c1 == '-') return 'I';
  if (c2 == '-') return 'D';
  return 'M';
}

/* parse_maf_seq_to_trim, cigar.rs:155-199 (group_by(cigar_cat_ext); zip truncates) */
void orc_parse_maf_seq_to_trim(const char* t, size_t tn, const char* q, size_t qn, uint64_t out[4]) {
  uint64_t head_ins = 0, head_del = 0, tail_ins = 0, tail_del = 0;
  int head_indel = 1;
  size_t cols = tn < qn ? tn : qn, i = 0;
  while (i < cols) {
    char k = cigar_cat_ext(t[i], q[i]);
    size_t j = i + 1;
    while (j < cols && cigar_cat_ext(t[j], q[j]) == k) j++;
    uint64_t count = (uint64_t)(j - i);
    if (k == 'I') {
      if (head_indel) head_ins += count;
      tail_ins = count;
    } else if (k == 'D') {
      if (head_indel) head_del += count;
      tail_del = count;
    } else { /* 'M' | 'X' | '=' */
      tail_ins = 0;
      tail_del = 0;
      head_indel = 0;
    }
    i = j;
  }
  out[0] = head_ins;
  out[1] = head_del;
  out[2] = tail_ins;
  out[3] = tail_del;
}

/* one record of converter::maf2chain (converter.rs:57-91): ChainHeader::try_from(&MAFRecord)
 * (chain.rs:103-140: target strand always '+', query coordinates through the strand-aware
 * accessors of maf.rs:433-450, then the same trim arithmetic as for PAF), its Display, the data
 * lines of parse_maf_seq_to_chain (cigar.rs:435-457: group_by(cigar_cat) through cigar_unit_chain)
 * and the closing "\n\n".  q_start / q_align are the s-line's own fields.  *out is malloc'd. */
void orc_maf2chain_record(const char* t_name, uint64_t t_size, uint64_t t_start, uint64_t t_align,
                          const char* q_name, uint64_t q_size, uint64_t q_sline_start,
                          uint64_t q_sline_align, int strand_neg, const char* t, size_t tn,
                          const char* q, size_t qn, uint64_t chain_id, char** out, size_t* out_len) {
  uint64_t tr[4];
  orc_parse_maf_seq_to_trim(t, tn, q, qn, tr);
  uint64_t qs = strand_neg ? q_size - q_sline_start - q_sline_align : q_sline_start;
  uint64_t qe = strand_neg ? q_size - q_sline_start : q_sline_start + q_sline_align;
  uint64_t ts = t_start, te = t_start + t_align;
  if (!strand_neg) {
    qs += tr[0];
    ts += tr[1];
    qe -= tr[2];
    te -= tr[3];
  } else {
    ts += tr[1];
    te -= tr[3];
    qs = q_size - (qe - tr[0]);
    qe = q_size - (qs + tr[2]);
  }
  sbuf sb = {NULL, 0, 0};
  char line[1024];
  snprintf(line, sizeof line, "chain\t255\t%s\t%llu\t+\t%llu\t%llu\t%s\t%llu\t%c\t%llu\t%llu\t%llu", t_name,
           (unsigned long long)t_size, (unsigned long long)ts, (unsigned long long)te, q_name,
           (unsigned long long)q_size, strand_neg ? '-' : '+', (unsigned long long)qs,
           (unsigned long long)qe, (unsigned long long)chain_id);
  sb_push(&sb, line, strlen(line));
  uint64_t size = 0, qd = 0, td = 0;
  size_t cols = tn < qn ? tn : qn, i = 0;
  while (i < cols) {
    char k = cigar_cat(t[i], q[i]);
    size_t j = i + 1;
    while (j < cols && cigar_cat(t[j], q[j]) == k) j++;
    uint64_t len = (uint64_t)(j - i);
    if (k == 'M') { /* cigar_unit_chain, cigar.rs:467-476 */
      if (size != 0 && td + qd != 0) {
        snprintf(line, sizeof line, "\n%llu\t%llu\t%llu", (unsigned long long)size, (unsigned long long)qd,
                 (unsigned long long)td);
        sb_push(&sb, line, strlen(line));
        size = 0;
      }
      size += len;
      td = 0;
      qd = 0;
    } else if (k == 'I') {
      td += len;
    } else {
      qd += len;
    }
    i = j;
  }
  snprintf(line, sizeof line, "\n%llu\n\n", (unsigned long long)size);
  sb_push(&sb, line, strlen(line));
  *out = sb.s;
  *out_len = sb.n;
}

/* parse_chain_to_cigar, cigar.rs:554-627.  lines = n_lines x (size, query_diff, target_diff) in the
 * order chain.rs:330-348 reads a data line.  *text = malloc'd NUL-terminated CIGAR (no tag). */
void orc_parse_chain_to_cigar(const uint64_t* lines, size_t n_lines, int strand_neg, orc_counts* out,
                              char** text) {
  memset(out, 0, sizeof(*out));
  sbuf sb = {0, 0, 0};
  sb_push(&sb, "", 0);
  if (strand_neg) out->inv_ev = 1;
  for (size_t k = 0; k < n_lines; k++) {
    const uint64_t match_len = lines[3 * k], del_len = lines[3 * k + 1], ins_len = lines[3 * k + 2];
    sb_printf_u64(&sb, match_len);
    sb_push(&sb, "M", 1);
    out->match += match_len;
    if (ins_len) {
      sb_printf_u64(&sb, ins_len);
      sb_push(&sb, "I", 1);
      if (strand_neg) {
        out->inv_ins_ev += 1;
        out->inv_ins_bp += ins_len;
      } else {
        out->ins_ev += 1;
        out->ins_bp += ins_len;
      }
    }
    if (del_len) {
      sb_printf_u64(&sb, del_len);
      sb_push(&sb, "D", 1);
      if (strand_neg) {
        out->inv_del_ev += 1;
        out->inv_del_bp += del_len;
      } else {
        out->del_ev += 1;
        out->del_bp += del_len;
      }
    }
  }
  sb_push(&sb, "\0", 1);
  *text = sb.s;
}

/* parse_chain_to_insert, converter.rs:360-388: String::insert_str of '-' runs at the running
 * column offset; returns ORC_PANIC where insert_str would panic (offset beyond the string).
 * *t / *q are malloc'd buffers, re-allocated as they grow. */
static int insert_dashes(char** s, size_t* n, uint64_t at, uint64_t count) {
  if (at > *n) return -1;
  char* r = (char*)malloc(*n + (size_t)count + 1);
  memcpy(r, *s, (size_t)at);
  memset(r + at, '-', (size_t)count);
  memcpy(r + at + count, *s + at, *n - (size_t)at);
  free(*s);
  *s = r;
  *n += (size_t)count;
  return 0;
}
int orc_parse_chain_to_insert(const uint64_t* lines, size_t n_lines, char** t, size_t* tn, char** q,
                              size_t* qn) {
  uint64_t cur = 0;
  for (size_t k = 0; k < n_lines; k++) {
    const uint64_t del_len = lines[3 * k + 1], ins_len = lines[3 * k + 2];
    cur += lines[3 * k];
    if (ins_len) {
      if (insert_dashes(t, tn, cur, ins_len)) return ORC_PANIC;
      cur += ins_len;
    }
    if (del_len) {
      if (insert_dashes(q, qn, cur, del_len)) return ORC_PANIC;
      cur += del_len;
    }
  }
  return ORC_OK;
}

/* ---- dotplot base-level segments ----------------------------------------- */
/* emit_baseplotdatas, cigar.rs:815-914, on a growing array of 5-u64 rows: ref_start, ref_end,
 * query_start, query_end, kind (0 'M', 1 'I', 2 'D'); reserve_query_start_end (:807-812) swaps the
 * query pair of a new row for '-' records, later updates go to the swapped field. */
typedef struct {
  uint64_t* v;
  size_t n, cap;
} segvec;
static void seg_push(segvec* sv, uint64_t rs, uint64_t re, uint64_t qs, uint64_t qe, uint64_t kind, int neg) {
  if (sv->n == sv->cap) {
    sv->cap = sv->cap ? sv->cap * 2 : 16;
    sv->v = (uint64_t*)realloc(sv->v, sv->cap * 5 * sizeof(uint64_t));
  }
  uint64_t* s = sv->v + 5 * sv->n++;
  s[0] = rs;
  s[1] = re;
  s[2] = neg ? qe : qs;
  s[3] = neg ? qs : qe;
  s[4] = kind;
}
static void emit_baseplotdatas(uint64_t* ref_off, uint64_t* q_off, int neg, char op, uint64_t length,
                               uint64_t cutoff, segvec* sv, int* last_m) {
  if (op == 'M' || op == '=' || op == 'X') {
    uint64_t ref_end = *ref_off + length, q_end = *q_off + length;
    if (!*last_m) {
      seg_push(sv, *ref_off, ref_end, *q_off, q_end, 0, neg);
    } else {
      uint64_t* m = sv->v + 5 * (sv->n - 1);
      m[1] = ref_end;
      if (neg) m[2] = q_end; else m[3] = q_end;
    }
    *ref_off += length;
    *q_off += length;
    *last_m = 1;
  } else if (op == 'I') {
    uint64_t q_end = *q_off + length;
    if (length > cutoff) {
      seg_push(sv, *ref_off, *ref_off, *q_off, q_end, 1, neg);
      *last_m = 0;
    } else if (*last_m) {
      uint64_t* m = sv->v + 5 * (sv->n - 1);
      if (neg) m[2] = q_end; else m[3] = q_end;
    }
    *q_off += length;
  } else if (op == 'D') {
    uint64_t ref_end = *ref_off + length;
    if (length > cutoff) {
      seg_push(sv, *ref_off, ref_end, *q_off, *q_off, 2, neg);
      *last_m = 0;
    } else if (*last_m) {
      sv->v[5 * (sv->n - 1) + 1] = ref_end;
    }
    *ref_off += length;
  }
}

/* parse_cigar_to_base_plotdata, cigar.rs:917-952; *segs = malloc'd n_segs x 5 u64 */
int orc_cigar_to_base_plotdata(const char* cg, size_t n, uint64_t t_start, uint64_t q_start, int strand_neg,
                               uint64_t cutoff, uint64_t** segs, size_t* n_segs, orc_err* err) {
  const char* p = cg;
  const char* end = cg + n;
  *segs = NULL;
  *n_segs = 0;
  if (strip_tag(&p, end, err)) return err->kind;
  segvec sv = {NULL, 0, 0};
  uint64_t r = t_start, q = q_start;
  int last_m = 0, any = 0;
  cst_t tok;
  while (parse_cigar_str_tuple(&p, end, &tok)) {
    const char* op;
    size_t op_n;
    uint64_t len;
    any = 1;
    if (cst2cu(&tok, &op, &op_n, &len, err)) {
      free(sv.v);
      return err->kind;
    }
    emit_baseplotdatas(&r, &q, strand_neg, op[0], len, cutoff, &sv, &last_m);
  }
  if (!any) { /* fold_many1 on an empty CIGAR */
    free(sv.v);
    return empty_cigar_panic(err);
  }
  *segs = sv.v;
  *n_segs = sv.n;
  return ORC_OK;
}

/* parse_maf_to_base_plotdata, cigar.rs:955-985 (group_by(cigar_cat_ext)) */
void orc_maf_to_base_plotdata(const char* t, size_t tn, const char* q, size_t qn, uint64_t t_start,
                              uint64_t q_start, int strand_neg, uint64_t cutoff, uint64_t** segs,
                              size_t* n_segs) {
  segvec sv = {NULL, 0, 0};
  uint64_t r = t_start, qo = q_start;
  int last_m = 0;
  size_t cols = tn < qn ? tn : qn, i = 0;
  while (i < cols) {
    char k = cigar_cat_ext(t[i], q[i]);
    size_t j = i + 1;
    while (j < cols && cigar_cat_ext(t[j], q[j]) == k) j++;
    emit_baseplotdatas(&r, &qo, strand_neg, k, (uint64_t)(j - i), cutoff, &sv, &last_m);
    i = j;
  }
  *segs = sv.v;
  *n_segs = sv.n;
}

/* per-record chunk loop of call_var_maf, caller.rs:115-149, with create_chunk_record
 * (:221-265: start += non-gap chars of the prefix, align_size = non-gap chars of the chunk) and
 * the strand-aware accessors of maf.rs:433-450,468-470 applied to the chunk record. */
int orc_call_var_maf_record(const char* chro, const char* q_chro, const char* t, const char* q,
                            size_t cols, uint64_t t_start, uint64_t t_size_unused,
                            uint64_t q_sline_start, uint64_t q_sline_align, uint64_t q_size,
                            int strand_neg, int if_snp, int if_inv, uint64_t svlen_cutoff,
                            size_t chunk_size, char** out, size_t* out_len) {
  (void)t_size_unused;
  (void)q_sline_align;
  size_t total = cols;
  size_t chunk_start = 0;
  while (chunk_start < total) {
    size_t safe_end =
        orc_find_safe_chunk_boundary(t, q, total, chunk_start, chunk_size, svlen_cutoff);
    /* create_chunk_record: naive prefix recount for every chunk (:240-251) */
    uint64_t nt_start = t_start, nq_start = q_sline_start, nt_align = 0, nq_align = 0;
    for (size_t i = 0; i < chunk_start; i++) {
      if (t[i] != '-') nt_start++;
      if (q[i] != '-') nq_start++;
    }
    for (size_t i = chunk_start; i < safe_end; i++) {
      if (t[i] != '-') nt_align++;
      if (q[i] != '-') nq_align++;
    }
    /* accessors on the chunk record */
    uint64_t c_t_start = nt_start;
    uint64_t c_t_end = nt_start + nt_align; /* maf.rs:468-470 */
    uint64_t c_q_start, c_q_end;
    if (strand_neg) { /* maf.rs:438-440, 448 */
      c_q_start = q_size - nq_start - nq_align;
      c_q_end = q_size - nq_start;
    } else {
      c_q_start = nq_start;
      c_q_end = nq_start + nq_align;
    }
    int rc = orc_call_within_var(chro, q_chro, t + chunk_start, q + chunk_start,
                                 safe_end - chunk_start, c_t_start, c_t_end, c_q_start, c_q_end,
                                 strand_neg, if_snp, svlen_cutoff, if_inv, out, out_len);
    if (rc) return rc;
    if (safe_end == chunk_start) break; /* guard: the reference would spin forever here */
    chunk_start = safe_end;
  }
  return ORC_OK;
}
