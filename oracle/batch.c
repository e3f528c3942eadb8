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
