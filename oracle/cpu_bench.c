/*
 * cpu_bench.c — the CPU baseline that bench.py reports next to the GPU number.  TEST INFRASTRUCTURE, part of
 * liboracle.so: only bench.py's cpu_baseline leg calls it.
 *
 * BASELINE.md section 3 / SURVEY.md 8(d): the reference (Rust) cannot be built here, so the baseline is this CPU
 * restatement in three forms over the SAME records the GPU step processes:
 *   mode 0  ref-faithful `stat`     per record orc_parse_paf_to_cigar (tokenise the text, cigar.rs:629-707); the
 *                                   reference runs this under rayon's par_bridge (stat.rs:67-81): `threads` workers
 *                                   pull records from one shared cursor, per-thread partial totals merged at the end
 *   mode 1  ref-faithful `paf2maf`  per record reverse_complement + parse_cigar_to_insert with String::insert_str's
 *                                   tail memmove per indel (cigar.rs:492-551, quadratic).  The reference's loop is
 *                                   serial (converter.rs:196); threads > 1 is "what a rayon-parallel paf2maf would get"
 *   mode 2  optimised stat+paf2maf  one pass over the packed u32 ops (no text, no quadratic inserts): counters and
 *                                   both gapped rows written left to right into a per-thread buffer
 * Every mode returns wall seconds of the parallel region, ops processed and a checksum (so nothing is optimised away).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "oracle.h"

typedef struct {
  double seconds;
  uint64_t ops, out_bytes, checksum;
} orc_bench_result;

typedef struct {
  int mode;
  uint32_t n;
  const char* cg_blob;
  const uint64_t* cg_off;
  const uint32_t* ops;
  const uint64_t* op_off;
  const uint8_t* strand;
  const uint8_t* t_pool;
  const uint64_t *t_off, *t_len;
  const uint8_t* q_pool;
  const uint64_t *q_off, *q_len;
  uint64_t cursor; /* shared: next record (par_bridge's single producer) */
  pthread_mutex_t mu;
  int failed;
} job_t;

typedef struct {
  job_t* job;
  uint64_t ops, out_bytes, checksum;
  orc_counts total;
} worker_t;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void add_counts(orc_counts* a, const orc_counts* b) {
  uint64_t* x = (uint64_t*)a;
  const uint64_t* y = (const uint64_t*)b;
  for (int k = 0; k < 11; k++) x[k] += y[k];
}

static const uint8_t* comp_lut(void) {
  static uint8_t lut[256];
  static int init = 0;
  if (!init) {
    for (int c = 0; c < 256; c++) lut[c] = (uint8_t)c;
    const char* a = "ACGTNacgtn";
    const char* b = "TGCANtgcan";
    for (int k = 0; a[k]; k++) lut[(unsigned char)a[k]] = (uint8_t)b[k];
    init = 1;
  }
  return lut;
}

/* mode 2: one record from packed ops; rows into buf (grown as needed) */
static int fast_record(const job_t* j, uint32_t i, uint8_t** buf, size_t* cap, worker_t* w) {
  const uint32_t* ops = j->ops + j->op_off[i];
  const size_t nop = (size_t)(j->op_off[i + 1] - j->op_off[i]);
  const uint8_t* t = j->t_pool + j->t_off[i];
  const uint8_t* q = j->q_pool + j->q_off[i];
  const size_t tl = (size_t)j->t_len[i], ql = (size_t)j->q_len[i];
  const int neg = j->strand[i] != 0;
  orc_counts c;
  memset(&c, 0, sizeof c);
  uint64_t L = 0;
  for (size_t k = 0; k < nop; k++) L += ops[k] >> 4; /* upper bound of the row length */
  if (2 * L + 64 > *cap) {
    *cap = 2 * L + 64 + *cap / 2;
    free(*buf);
    *buf = (uint8_t*)malloc(*cap);
    if (!*buf) return 1;
  }
  uint8_t* tr = *buf;
  uint8_t* qr = *buf + L + 32;
  const uint8_t* lut = comp_lut();
  size_t x = 0, tp = 0, qp = 0;
  for (size_t k = 0; k < nop; k++) {
    const uint32_t code = ops[k] & 15u;
    const size_t len = ops[k] >> 4;
    switch (code) {
      case 0: case 7: case 8: /* M = X */
        if (tp + len > tl || qp + len > ql) return 2;
        if (code == 8) c.mismatch += len; else c.match += len;
        memcpy(tr + x, t + tp, len);
        if (!neg) memcpy(qr + x, q + qp, len);
        else for (size_t b = 0; b < len; b++) qr[x + b] = lut[q[ql - 1 - (qp + b)]];
        tp += len; qp += len; x += len;
        break;
      case 1: case 9: /* I (9: continuation of a split op) */
        if (qp + len > ql) return 2;
        if (neg) { c.inv_ins_bp += len; c.inv_ins_ev += code == 1; } else { c.ins_bp += len; c.ins_ev += code == 1; }
        memset(tr + x, '-', len);
        if (!neg) memcpy(qr + x, q + qp, len);
        else for (size_t b = 0; b < len; b++) qr[x + b] = lut[q[ql - 1 - (qp + b)]];
        qp += len; x += len;
        break;
      case 2: case 10: /* D */
        if (tp + len > tl) return 2;
        if (neg) { c.inv_del_bp += len; c.inv_del_ev += code == 2; } else { c.del_bp += len; c.del_ev += code == 2; }
        memcpy(tr + x, t + tp, len);
        memset(qr + x, '-', len);
        tp += len; x += len;
        break;
      default: return 3;
    }
  }
  c.inv_ev = neg ? 1u : 0u;
  add_counts(&w->total, &c);
  w->ops += nop;
  w->out_bytes += 2 * x;
  if (x) w->checksum += (uint64_t)tr[0] + tr[x - 1] + qr[0] + qr[x - 1] + tr[x / 2] * 3u + qr[x / 3];
  return 0;
}

static void* worker_main(void* arg) {
  worker_t* w = (worker_t*)arg;
  job_t* j = w->job;
  uint8_t* buf = NULL;
  size_t cap = 0;
  for (;;) {
    /* par_bridge hands out items one by one from a locked iterator; a small block keeps the lock off the profile */
    pthread_mutex_lock(&j->mu);
    uint64_t a = j->cursor;
    uint64_t b = a + 4 < j->n ? a + 4 : j->n;
    j->cursor = b;
    pthread_mutex_unlock(&j->mu);
    if (a >= j->n) break;
    for (uint64_t i = a; i < b; i++) {
      const char* cg = j->cg_blob ? j->cg_blob + j->cg_off[i] : NULL;
      const size_t cgn = j->cg_blob ? (size_t)(j->cg_off[i + 1] - j->cg_off[i]) : 0;
      orc_err err;
      if (j->mode == 0) {
        orc_counts c;
        if (orc_parse_paf_to_cigar(cg, cgn, j->strand[i], &c, &err)) { j->failed = 1; continue; }
        add_counts(&w->total, &c);
        w->ops += (uint64_t)(j->op_off[i + 1] - j->op_off[i]);
        w->checksum += c.match + 3 * c.mismatch;
      } else if (j->mode == 1) {
        size_t tn = (size_t)j->t_len[i], qn = (size_t)j->q_len[i];
        char* t = (char*)malloc(tn + 1);
        char* q = (char*)malloc(qn + 1);
        if (!t || !q) { j->failed = 1; free(t); free(q); continue; }
        memcpy(t, j->t_pool + j->t_off[i], tn); /* the faidx fetch returns an owned String (converter.rs:219-225) */
        if (j->strand[i]) {
          if (orc_reverse_complement((const char*)(j->q_pool + j->q_off[i]), qn, q, &err)) j->failed = 1;
        } else {
          memcpy(q, j->q_pool + j->q_off[i], qn);
        }
        if (orc_parse_cigar_to_insert(cg, cgn, &t, &tn, &q, &qn, &err)) j->failed = 1;
        w->ops += (uint64_t)(j->op_off[i + 1] - j->op_off[i]);
        w->out_bytes += tn + qn;
        if (tn && qn) w->checksum += (uint64_t)(unsigned char)t[0] + (unsigned char)t[tn - 1] + (unsigned char)q[0] + (unsigned char)q[qn - 1];
        free(t);
        free(q);
      } else {
        if (fast_record(j, (uint32_t)i, &buf, &cap, w)) j->failed = 1;
      }
    }
  }
  free(buf);
  return NULL;
}

int orc_bench_run(int mode, int threads, uint32_t n, const char* cg_blob, const uint64_t* cg_off, const uint32_t* ops,
                  const uint64_t* op_off, const uint8_t* strand, const uint8_t* t_pool, const uint64_t* t_off,
                  const uint64_t* t_len, const uint8_t* q_pool, const uint64_t* q_off, const uint64_t* q_len,
                  orc_bench_result* out) {
  if (threads < 1) threads = 1;
  job_t j;
  memset(&j, 0, sizeof j);
  j.mode = mode;
  j.n = n;
  j.cg_blob = cg_blob;
  j.cg_off = cg_off;
  j.ops = ops;
  j.op_off = op_off;
  j.strand = strand;
  j.t_pool = t_pool;
  j.t_off = t_off;
  j.t_len = t_len;
  j.q_pool = q_pool;
  j.q_off = q_off;
  j.q_len = q_len;
  pthread_mutex_init(&j.mu, NULL);
  (void)comp_lut();
  worker_t* ws = (worker_t*)calloc((size_t)threads, sizeof(worker_t));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  if (!ws || !th) return 1;
  const double t0 = now_s();
  for (int k = 0; k < threads; k++) {
    ws[k].job = &j;
    if (threads == 1)
      worker_main(&ws[k]); /* -t 1: the calling thread */
    else
      pthread_create(&th[k], NULL, worker_main, &ws[k]);
  }
  if (threads > 1)
    for (int k = 0; k < threads; k++) pthread_join(th[k], NULL);
  /* merge of the per-thread partials (stat.rs:167-223 / pafcov.rs:29-53 do the same) */
  orc_counts total;
  memset(&total, 0, sizeof total);
  memset(out, 0, sizeof *out);
  for (int k = 0; k < threads; k++) {
    add_counts(&total, &ws[k].total);
    out->ops += ws[k].ops;
    out->out_bytes += ws[k].out_bytes;
    out->checksum += ws[k].checksum;
  }
  out->seconds = now_s() - t0;
  out->checksum += total.match + total.mismatch + total.ins_bp + total.del_bp + total.inv_ins_bp + total.inv_del_bp;
  free(ws);
  free(th);
  pthread_mutex_destroy(&j.mu);
  return j.failed;
}
