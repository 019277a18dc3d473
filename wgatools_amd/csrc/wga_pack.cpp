/*
 * wga_pack.cpp — host side of the boundary: CIGAR text -> packed u32 ops.
 *
 * Replaces the nom tokeniser every reference consumer re-runs per record
 * (parse_cigar_str_tuple cigar.rs:59-75, cst2cu cigar.rs:43-56, parse_str2u64 utils.rs:69-74):
 *   length = maximal leading run of ASCII digits (may be empty  -> ParseIntError(""))
 *   op     = maximal following run of non-digits; must be exactly one char
 *            (empty or longer -> CigarOpInvalid(token)); the op is checked before the length.
 * Tokenising happens once; all kernels consume the packed stream.
 */
#include <stddef.h>
#include <stdint.h>

#include "../../include/wga_hip.h"

static inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
static inline size_t utf8_len(unsigned char c) {
  if (c < 0x80) return 1;
  if ((c >> 5) == 0x6) return 2;
  if ((c >> 4) == 0xE) return 3;
  if ((c >> 3) == 0x1E) return 4;
  return 1;
}

static inline uint32_t op_code(char c) {
  switch (c) {
    case 'M': return WGA_OP_M;
    case 'I': return WGA_OP_I;
    case 'D': return WGA_OP_D;
    case 'N': return WGA_OP_N;
    case 'S': return WGA_OP_S;
    case 'H': return WGA_OP_H;
    case 'P': return WGA_OP_P;
    case '=': return WGA_OP_EQ;
    case 'X': return WGA_OP_X;
    default: return WGA_OP_OTHER;
  }
}

extern "C" int wga_cigar_pack(const char* text, size_t len, uint32_t* ops, size_t cap,
                              size_t* n_ops, int32_t* err, size_t* err_tok_off,
                              size_t* err_tok_len) {
  size_t n = 0;
  int32_t e = WGA_REC_OK;
  size_t eoff = 0, elen = 0;
  if (len == 0) {
    /* fold_many1 on an empty CIGAR -> Many1 error -> errors.rs:92 slices input[..10]: panic */
    e = WGA_REC_PANIC;
  }
  size_t p = 0;
  while (p < len) {
    size_t ls = p;
    while (p < len && is_digit(text[p])) p++;
    size_t ln = p - ls;
    size_t os = p;
    while (p < len && !is_digit(text[p])) p++;
    size_t on = p - os;
    if (on == 0 || utf8_len((unsigned char)text[os]) < on) { /* cigar.rs:46-53 */
      e = WGA_REC_CIGAR_OP_INVALID;
      eoff = os;
      elen = on;
      break;
    }
    if (ln == 0) { /* "".parse::<u64>() */
      e = WGA_REC_PARSE_INT;
      eoff = ls;
      elen = 0;
      break;
    }
    uint64_t v = 0;
    bool ovf = false;
    for (size_t i = 0; i < ln; i++) {
      uint64_t d = (uint64_t)(text[ls + i] - '0');
      if (v > (UINT64_MAX - d) / 10) {
        ovf = true;
        break;
      }
      v = v * 10 + d;
    }
    if (ovf) {
      e = WGA_REC_PARSE_INT;
      eoff = ls;
      elen = ln;
      break;
    }
    uint32_t code = on == 1 ? op_code(text[os]) : (uint32_t)WGA_OP_OTHER;
    /* lengths >= 2^28 are split; later pieces of an I / D carry a continuation code so that
     * ins_event / del_event stay one per op (cigar.rs:667-684) */
    uint32_t cont = code == WGA_OP_I ? (uint32_t)WGA_OP_I_CONT
                                     : code == WGA_OP_D ? (uint32_t)WGA_OP_D_CONT : code;
    bool first = true;
    do {
      uint64_t piece = v > WGA_OP_MAX_LEN ? WGA_OP_MAX_LEN : v;
      if (n < cap && ops) ops[n] = WGA_PACK_OP(piece, first ? code : cont);
      n++;
      v -= piece;
      first = false;
    } while (v > 0);
  }
  if (n_ops) *n_ops = n;
  if (err) *err = e;
  if (err_tok_off) *err_tok_off = eoff;
  if (err_tok_len) *err_tok_len = elen;
  return (n > cap || (n && !ops)) ? WGA_E_TOO_SMALL : WGA_OK;
}

extern "C" size_t wga_cigar_pack_bound(const char* text, size_t len) {
  size_t n = 0;
  (void)wga_cigar_pack(text, len, nullptr, 0, &n, nullptr, nullptr, nullptr);
  return n;
}
