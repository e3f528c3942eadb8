/* ORACLE — test infrastructure only.  Batch (SoA) driver over the per-item restatement; this is the
 * timed "restated CPU path" of bench.py's cpu_baseline leg (kind "port").  pthreads, static split. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "afx_oracle.h"

typedef struct {
  const afxo_ctx* c;
  const afx_shape* shape;
  const afx_presentation_soa* b;
  size_t count, lo, hi;
  uint8_t* status;
} job_t;

static void gather(afxo_presentation* p, const afx_shape* s, const afx_presentation_soa* b, size_t count, size_t i) {
  memset(p, 0, sizeof *p);
  p->n_attributes = s->n_attributes;
  p->n_responses = s->n_responses;
  memcpy(p->challenge, b->challenge + 32 * i, 32);
  for (uint32_t k = 0; k < s->n_responses && k < 3 + AFX_MAX_ATTRIBUTES; k++) memcpy(p->responses[k], b->responses + 32 * (k * count + i), 32);
  memcpy(p->C_x_0, b->C_x_0 + 32 * i, 32);
  memcpy(p->C_x_1, b->C_x_1 + 32 * i, 32);
  memcpy(p->C_V, b->C_V + 32 * i, 32);
  for (uint32_t k = 0; k < s->n_attributes; k++) {
    memcpy(p->C_y[k], b->C_y + 32 * (k * count + i), 32);
    p->kinds[k] = s->kinds[k];
    if (s->kinds[k] == AFX_ENC_PUBLIC_SCALAR || s->kinds[k] == AFX_ENC_PUBLIC_POINT)
      memcpy(p->attr_values[k], b->attr_values + 32 * (k * count + i), 32);
  }
  p->n_hidden_scalars = s->n_hidden_scalars;
  memcpy(p->hidden_scalar_indices, s->hidden_scalar_indices, sizeof s->hidden_scalar_indices);
  p->n_enc_proofs = s->n_enc_proofs;
  for (uint32_t e = 0; e < s->n_enc_proofs; e++) {
    const afx_encproof_soa* q = &b->enc[e];
    afxo_encproof* o = &p->enc[e];
    memcpy(o->challenge, q->challenge + 32 * i, 32);
    for (int k = 0; k < 6; k++) memcpy(o->responses[k], q->responses + 32 * (k * count + i), 32);
    memcpy(o->pk, q->pk + 32 * i, 32);
    memcpy(o->E1, q->E1 + 32 * i, 32);
    memcpy(o->E2, q->E2 + 32 * i, 32);
    memcpy(o->C_y_1, q->C_y_1 + 32 * i, 32);
    memcpy(o->C_y_2, q->C_y_2 + 32 * i, 32);
    memcpy(o->C_y_3, q->C_y_3 + 32 * i, 32);
    memcpy(o->C_y_2p, q->C_y_2p + 32 * i, 32);
    o->index = s->enc_indices[e];
  }
}

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  afxo_presentation* p = (afxo_presentation*)malloc(sizeof *p);
  for (size_t i = j->lo; i < j->hi; i++) {
    gather(p, j->shape, j->b, j->count, i);
    int r = afxo_verify_presentation(j->c, p);
    j->status[i] = (uint8_t)(r == 0 ? AFX_ST_OK : AFX_ST_VERIFICATION_FAILURE);
  }
  free(p);
  return NULL;
}

int afxo_verify_presentations_soa(const afxo_ctx* c, const afx_shape* shape, const afx_presentation_soa* batch, size_t count,
                                  uint8_t* status, int threads) {
  if (shape->n_attributes > AFX_MAX_ATTRIBUTES || shape->n_responses > 3 + AFX_MAX_ATTRIBUTES ||
      shape->n_hidden_scalars > AFX_MAX_ATTRIBUTES || shape->n_enc_proofs > AFX_MAX_ATTRIBUTES) {
    memset(status, AFX_ST_VERIFICATION_FAILURE, count);
    return 0;
  }
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t th[256];
  job_t jobs[256];
  for (int t = 0; t < threads; t++) {
    jobs[t] = (job_t){ c, shape, batch, count, count * (size_t)t / (size_t)threads, count * (size_t)(t + 1) / (size_t)threads, status };
    if (threads == 1) worker(&jobs[t]);
    else pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  if (threads > 1)
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  return 0;
}

/* ---- batch forms for the full-size parity tests (tests/test_gpu_full_size.py, SURVEY.md section 8d: "every recomputed commitment +
 * challenge compared ... on a 2^16 sample") ------------------------------------------------------------------------------------
 * Beside the status, every proof's verifier reports the challenge it RECOMPUTES (zkp verify_compact's c', a hash over all of
 * its recomputed commitments): trace[(r * count + i) * 32], r = 0 the presentation proof, 1 + e the e-th attached proof of
 * encryption; reached[r * count + i] = 1 when that verifier got as far as its commitments (an item rejected before the
 * transcript stage has no challenge).  Each proof is verified on its own for the trace (the reference stops at the first
 * failing proof, presentation.rs:435-440; the engine computes them all); the status is afxo_verify_presentation's. */
typedef struct {
  job_t j;
  uint8_t* trace;
  uint8_t* reached;
} tjob_t;

static void* traced_worker(void* arg) {
  tjob_t* t = (tjob_t*)arg;
  job_t* j = &t->j;
  afxo_presentation* p = (afxo_presentation*)malloc(sizeof *p);
  const uint32_t ne = j->shape->n_enc_proofs;
  uint8_t commits[64 * 32];
  for (size_t i = j->lo; i < j->hi; i++) {
    gather(p, j->shape, j->b, j->count, i);
    j->status[i] = (uint8_t)(afxo_verify_presentation(j->c, p) == 0 ? AFX_ST_OK : AFX_ST_VERIFICATION_FAILURE);
    int nc = 0;
    p->n_enc_proofs = 0;
    afxo_debug_reset();
    (void)afxo_verify_presentation(j->c, p);
    afxo_debug_last(commits, &nc, t->trace + 32 * i);
    t->reached[i] = nc > 0;
    for (uint32_t e = 0; e < ne; e++) {
      afxo_debug_reset();
      (void)afxo_verify_encryption_proof(j->c, &p->enc[e]);
      afxo_debug_last(commits, &nc, t->trace + 32 * ((size_t)(1 + e) * j->count + i));
      t->reached[(size_t)(1 + e) * j->count + i] = nc > 0;
    }
  }
  free(p);
  return NULL;
}

int afxo_verify_presentations_soa_traced(const afxo_ctx* c, const afx_shape* shape, const afx_presentation_soa* batch, size_t count,
                                         uint8_t* status, uint8_t* trace, uint8_t* reached, int threads) {
  if (shape->n_attributes > AFX_MAX_ATTRIBUTES || shape->n_responses > 3 + AFX_MAX_ATTRIBUTES ||
      shape->n_hidden_scalars > AFX_MAX_ATTRIBUTES || shape->n_enc_proofs > AFX_MAX_ATTRIBUTES)
    return -1;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t th[256];
  tjob_t jobs[256];
  for (int t = 0; t < threads; t++) {
    jobs[t].j = (job_t){ c, shape, batch, count, count * (size_t)t / (size_t)threads, count * (size_t)(t + 1) / (size_t)threads, status };
    jobs[t].trace = trace;
    jobs[t].reached = reached;
    if (threads == 1) traced_worker(&jobs[t]);
    else pthread_create(&th[t], NULL, traced_worker, &jobs[t]);
  }
  if (threads > 1)
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  return 0;
}

/* Issuer::issue and CredentialIssuance::verify over struct-of-arrays batches (values [n][count][32]: the scalar, the point, or a
 * plaintext's M1 - all that the tag and the issuance proof read of an attribute; the same layout as afx_attributes_soa). */
typedef struct {
  const afxo_ctx* c;
  uint32_t n, nr;
  const uint8_t* kinds;
  const uint8_t *values, *t_wide, *U_wide, *seed;
  uint8_t *t, *U, *V, *challenge, *responses, *status, *trace, *reached;
  size_t count, lo, hi;
  int verify;
} ijob_t;

static void* issuance_worker(void* arg) {
  ijob_t* j = (ijob_t*)arg;
  uint8_t vals[AFX_MAX_ATTRIBUTES * 96], resp[(AFX_MAX_ATTRIBUTES + 5) * 32], commits[64 * 32];
  for (size_t i = j->lo; i < j->hi; i++) {
    memset(vals, 0, sizeof vals);
    for (uint32_t k = 0; k < j->n; k++) memcpy(vals + 96 * k, j->values + 32 * ((size_t)k * j->count + i), 32);
    if (!j->verify) {
      const int st = afxo_issue(j->c, j->n, j->kinds, vals, j->t_wide + 64 * i, j->U_wide + 64 * i, j->seed + 32 * i, j->t + 32 * i,
                                j->U + 32 * i, j->V + 32 * i, j->challenge + 32 * i, resp);
      j->status[i] = (uint8_t)st;
      for (uint32_t k = 0; k < j->nr; k++) memcpy(j->responses + 32 * ((size_t)k * j->count + i), resp + 32 * k, 32);
    } else {
      for (uint32_t k = 0; k < j->nr; k++) memcpy(resp + 32 * k, j->responses + 32 * ((size_t)k * j->count + i), 32);
      int nc = 0;
      afxo_debug_reset();
      const int st = afxo_issuance_verify(j->c, j->n, j->kinds, vals, j->t + 32 * i, j->U + 32 * i, j->V + 32 * i, j->challenge + 32 * i, resp, j->nr);
      j->status[i] = (uint8_t)(st == 0 ? AFX_ST_OK : AFX_ST_VERIFICATION_FAILURE);
      if (j->trace) { afxo_debug_last(commits, &nc, j->trace + 32 * i); j->reached[i] = nc > 0; }
    }
  }
  return NULL;
}

static int issuance_batch(ijob_t proto, int threads) {
  if (proto.n > AFX_MAX_ATTRIBUTES || proto.nr > AFX_MAX_ATTRIBUTES + 5) return -1;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t th[256];
  ijob_t jobs[256];
  for (int t = 0; t < threads; t++) {
    jobs[t] = proto;
    jobs[t].lo = proto.count * (size_t)t / (size_t)threads;
    jobs[t].hi = proto.count * (size_t)(t + 1) / (size_t)threads;
    if (threads == 1) issuance_worker(&jobs[t]);
    else pthread_create(&th[t], NULL, issuance_worker, &jobs[t]);
  }
  if (threads > 1)
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  return 0;
}

int afxo_issue_soa(const afxo_ctx* c, uint32_t n, const uint8_t* kinds, const uint8_t* values, const uint8_t* t_wide, const uint8_t* U_wide,
                   const uint8_t* seed, size_t count, uint8_t* t, uint8_t* U, uint8_t* V, uint8_t* challenge, uint8_t* responses,
                   uint8_t* status, int threads) {
  ijob_t p;
  memset(&p, 0, sizeof p);
  p.c = c; p.n = n; p.nr = afxo_ctx_n(c) + 5; p.kinds = kinds; p.values = values; p.t_wide = t_wide; p.U_wide = U_wide; p.seed = seed;
  p.t = t; p.U = U; p.V = V; p.challenge = challenge; p.responses = responses; p.status = status; p.count = count; p.verify = 0;
  return issuance_batch(p, threads);
}

/* trace[count][32] / reached[count]: the challenge CredentialIssuance::verify recomputes (issuance.rs:217); may both be null */
int afxo_verify_issuances_soa_traced(const afxo_ctx* c, uint32_t n, const uint8_t* kinds, const uint8_t* values, const uint8_t* t,
                                     const uint8_t* U, const uint8_t* V, const uint8_t* challenge, const uint8_t* responses, uint32_t nr,
                                     size_t count, uint8_t* status, uint8_t* trace, uint8_t* reached, int threads) {
  ijob_t p;
  memset(&p, 0, sizeof p);
  p.c = c; p.n = n; p.nr = nr; p.kinds = kinds; p.values = values;
  p.t = (uint8_t*)t; p.U = (uint8_t*)U; p.V = (uint8_t*)V; p.challenge = (uint8_t*)challenge; p.responses = (uint8_t*)responses;
  p.status = status; p.trace = trace; p.reached = reached; p.count = count; p.verify = 1;
  return issuance_batch(p, threads);
}
