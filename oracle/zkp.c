/* ORACLE — test infrastructure only (see afx_oracle_internal.h header).
 *
 * zkp 0.7 `toolbox::{prover::Prover, verifier::Verifier, SchnorrCS}` + `CompactProof` [3P, not under
 * /root/reference], restated per SURVEY.md App. A.2.  Reference call sites:
 * src/nizk/presentation.rs:187-284,355-435; src/nizk/encryption.rs:81-130,160-209;
 * src/nizk/issuance.rs:48-128,142-217.
 */
#include "afx_oracle_internal.h"

__thread int zkp_debug_ncommit;
__thread uint8_t zkp_debug_commit[ZKP_MAX_CONSTRAINTS][32];
__thread uint8_t zkp_debug_challenge[32];

static void append(zkp_cs* z, const char* label, const uint8_t* msg, size_t mlen) {
  merlin_append_message(&z->t, (const uint8_t*)label, strlen(label), msg, mlen);
}

void zkp_init(zkp_cs* z, int is_prover, const char* transcript_label, const char* proof_label) {
  afxo_init_constants();
  memset(z, 0, sizeof *z);
  z->is_prover = is_prover;
  merlin_new(&z->t, (const uint8_t*)transcript_label, strlen(transcript_label));
  /* TranscriptProtocol::domain_sep */
  append(z, "dom-sep", (const uint8_t*)"schnorrzkp/1.0/ristretto255", 27);
  append(z, "dom-sep", (const uint8_t*)proof_label, strlen(proof_label));
}

int zkp_alloc_scalar(zkp_cs* z, const char* label, const sc* value) {
  append(z, "scvar", (const uint8_t*)label, strlen(label));
  if (value) z->scalars[z->n_scalars] = *value;
  return z->n_scalars++;
}

int zkp_alloc_point_prover(zkp_cs* z, const char* label, const ge* p) {
  int i = z->n_points++;
  z->points[i] = *p;
  z->point_labels[i] = label;
  ristretto_encode(z->enc[i], p);
  append(z, "ptvar", (const uint8_t*)label, strlen(label));
  append(z, "val", z->enc[i], 32);
  return i;
}

int zkp_alloc_point_verifier(zkp_cs* z, const char* label, const uint8_t enc[32]) {
  int i = z->n_points++;
  z->point_labels[i] = label;
  memcpy(z->enc[i], enc, 32);
  /* validate_and_append_point_var: identity encoding is rejected */
  uint8_t acc = 0;
  for (int k = 0; k < 32; k++) acc |= enc[k];
  if (acc == 0) z->failed = 1;
  append(z, "ptvar", (const uint8_t*)label, strlen(label));
  append(z, "val", enc, 32);
  return i;
}

void zkp_constrain(zkp_cs* z, int lhs, int n, const int* scs, const int* pts) {
  zkp_constraint* c = &z->cs[z->n_constraints++];
  c->lhs = lhs;
  c->n = n;
  for (int i = 0; i < n; i++) { c->sc[i] = scs[i]; c->pt[i] = pts[i]; }
}

static void get_challenge(zkp_cs* z, sc* c) {
  uint8_t wide[64];
  merlin_challenge_bytes(&z->t, (const uint8_t*)"chal", 4, wide, 64);
  sc_reduce_wide(c, wide);
}

void zkp_prove_compact(zkp_cs* z, const uint8_t rng_seed[32], sc* challenge, sc* responses) {
  /* TranscriptRngBuilder: clone the strobe, rekey with every witness, finalize with 32 external bytes */
  strobe128 rng = z->t.s;
  static const uint8_t len32[4] = { 32, 0, 0, 0 };
  for (int i = 0; i < z->n_scalars; i++) {
    strobe_meta_ad(&rng, (const uint8_t*)"", 0, 0);
    strobe_meta_ad(&rng, len32, 4, 1);
    strobe_key(&rng, z->scalars[i].b, 32, 0);
  }
  strobe_meta_ad(&rng, (const uint8_t*)"rng", 3, 0);
  strobe_key(&rng, rng_seed, 32, 0);
  sc blind[ZKP_MAX_SCALARS];
  static const uint8_t len64[4] = { 64, 0, 0, 0 };
  for (int i = 0; i < z->n_scalars; i++) {
    uint8_t wide[64];
    strobe_meta_ad(&rng, len64, 4, 0);
    strobe_prf(&rng, wide, 64, 0);
    sc_reduce_wide(&blind[i], wide);
  }
  zkp_debug_ncommit = z->n_constraints;
  for (int j = 0; j < z->n_constraints; j++) {
    const zkp_constraint* c = &z->cs[j];
    sc s[ZKP_MAX_TERMS];
    ge p[ZKP_MAX_TERMS];
    for (int k = 0; k < c->n; k++) { s[k] = blind[c->sc[k]]; p[k] = z->points[c->pt[k]]; }
    ge R;
    ge_multiscalar(&R, s, p, c->n);
    uint8_t enc[32];
    ristretto_encode(enc, &R);
    memcpy(zkp_debug_commit[j], enc, 32);
    append(z, "blindcom", (const uint8_t*)z->point_labels[c->lhs], strlen(z->point_labels[c->lhs]));
    append(z, "val", enc, 32);
  }
  get_challenge(z, challenge);
  memcpy(zkp_debug_challenge, challenge->b, 32);
  for (int i = 0; i < z->n_scalars; i++) sc_muladd(&responses[i], &z->scalars[i], challenge, &blind[i]);
}

int zkp_verify_compact(zkp_cs* z, const uint8_t challenge[32], const uint8_t* responses, int n_responses) {
  if (z->failed) return 0;
  if (n_responses != z->n_scalars) return 0;
  for (int i = 0; i < z->n_points; i++)
    if (!ristretto_decode(&z->points[i], z->enc[i])) return 0;
  sc c, minus_c;
  memcpy(c.b, challenge, 32);
  sc_neg(&minus_c, &c);
  zkp_debug_ncommit = z->n_constraints;
  for (int j = 0; j < z->n_constraints; j++) {
    const zkp_constraint* cn = &z->cs[j];
    sc s[ZKP_MAX_TERMS];
    ge p[ZKP_MAX_TERMS];
    for (int k = 0; k < cn->n; k++) {
      memcpy(s[k].b, responses + 32 * cn->sc[k], 32);
      p[k] = z->points[cn->pt[k]];
    }
    s[cn->n] = minus_c;
    p[cn->n] = z->points[cn->lhs];
    ge R;
    ge_multiscalar_vartime(&R, s, p, cn->n + 1);
    uint8_t enc[32];
    ristretto_encode(enc, &R);
    memcpy(zkp_debug_commit[j], enc, 32);
    append(z, "blindcom", (const uint8_t*)z->point_labels[cn->lhs], strlen(z->point_labels[cn->lhs]));
    append(z, "val", enc, 32);
  }
  sc c2;
  get_challenge(z, &c2);
  memcpy(zkp_debug_challenge, c2.b, 32);
  return memcmp(c2.b, challenge, 32) == 0;
}
