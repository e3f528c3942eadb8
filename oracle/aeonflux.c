/* ORACLE — test infrastructure only (see afx_oracle_internal.h header).  PARITY UNPINNED by the
 * reference (no golden vectors there); pinned by third-party KATs and libsodium (tests/golden/).
 *
 * CPU restatement of aeonflux's statements, following the reference line by line in behaviour:
 *   setup       src/parameters.rs:196-362, src/amacs.rs:89-125
 *   issue       src/issuer.rs:111-124, src/amacs.rs:225-294, src/nizk/issuance.rs:40-129
 *   issuance ok src/issuer.rs:48-57, src/nizk/issuance.rs:132-218
 *   show        src/credential.rs:37-46, src/nizk/presentation.rs:139-321, src/nizk/encryption.rs:58-142,
 *               src/symmetric.rs:252-261
 *   verify      src/issuer.rs:141-147, src/nizk/presentation.rs:324-443, src/nizk/encryption.rs:154-210
 *   symmetric   src/symmetric.rs:135-143,197-289; src/encoding.rs:56-82
 */
#include <stdlib.h>
#include "afx_oracle.h"
#include "afx_oracle_internal.h"

struct afxo_ctx {
  uint32_t n, g;
  ge G, G_w, G_wp, G_x0, G_x1, G_y[AFX_MAX_ATTRIBUTES], G_m[AFX_MAX_ATTRIBUTES], G_V, G_a, G_a0, G_a1;
  int has_key;
  sc w, wp, x0, x1, y[AFX_MAX_ATTRIBUTES];
  ge W;
  int has_issuer_params;
  ge C_W, I;
  int strict; /* SURVEY.md section 8f rank 4 (NOT the reference's behaviour): see afxo_ctx_set_strict */
};

static uint32_t rd32(const uint8_t* b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }
static void wr32(uint8_t* b, uint32_t x) { b[0] = (uint8_t)x; b[1] = (uint8_t)(x >> 8); b[2] = (uint8_t)(x >> 16); b[3] = (uint8_t)(x >> 24); }

/* src/parameters.rs:34-40 */
size_t afxo_sizeof_system_parameters(uint32_t n) {
  if (n < 3) return 32 * (5 + 3 + (size_t)n + 4) + 4;
  return 32 * (5 + 2 * (size_t)n + 4) + 4;
}
/* src/amacs.rs:44-46 */
size_t afxo_sizeof_secret_key(uint32_t n) { return 32 * (5 + (size_t)n) + 4; }

/* SystemParameters::from_bytes, src/parameters.rs:92-153 */
static int parse_params(struct afxo_ctx* c, const uint8_t* b, size_t len) {
  if (len < 4) return 0;
  uint32_t n = rd32(b);
  if (n == 0 || n > AFX_MAX_ATTRIBUTES) return 0;
  if (len != afxo_sizeof_system_parameters(n)) return 0;
  c->n = n;
  c->g = n < 3 ? 3 : n;
  const uint8_t* p = b + 4;
#define NEXT(dst) do { if (!ristretto_decode(&(dst), p)) return 0; p += 32; } while (0)
  NEXT(c->G); NEXT(c->G_w); NEXT(c->G_wp); NEXT(c->G_x0); NEXT(c->G_x1);
  for (uint32_t i = 0; i < c->g; i++) NEXT(c->G_y[i]);
  for (uint32_t i = 0; i < c->n; i++) NEXT(c->G_m[i]);
  NEXT(c->G_V); NEXT(c->G_a); NEXT(c->G_a0); NEXT(c->G_a1);
#undef NEXT
  return 1;
}

/* amacs::SecretKey::from_bytes as intended (every y_i read; src/amacs.rs:128-155) */
static int parse_key(struct afxo_ctx* c, const uint8_t* b, size_t len) {
  if (len < 4) return 0;
  uint32_t n = rd32(b);
  if (n != c->n || len != afxo_sizeof_secret_key(n)) return 0;
  const uint8_t* p = b + 4;
#define NEXTS(dst) do { if (!sc_is_canonical(p)) return 0; memcpy((dst).b, p, 32); p += 32; } while (0)
  NEXTS(c->w); NEXTS(c->wp); NEXTS(c->x0); NEXTS(c->x1);
  for (uint32_t i = 0; i < n; i++) NEXTS(c->y[i]);
#undef NEXTS
  if (!ristretto_decode(&c->W, p)) return 0;
  c->has_key = 1;
  return 1;
}

afxo_ctx* afxo_ctx_new(const uint8_t* params, size_t plen, const uint8_t* key, size_t klen, const uint8_t* issuer_params) {
  afxo_init_constants();
  struct afxo_ctx* c = (struct afxo_ctx*)calloc(1, sizeof *c);
  if (!c) return NULL;
  if (!parse_params(c, params, plen)) { free(c); return NULL; }
  if (key && klen && !parse_key(c, key, klen)) { free(c); return NULL; }
  if (issuer_params) {
    if (!ristretto_decode(&c->C_W, issuer_params) || !ristretto_decode(&c->I, issuer_params + 32)) { free(c); return NULL; }
    c->has_issuer_params = 1;
  }
  return c;
}
void afxo_ctx_free(afxo_ctx* c) {
  if (c) { memset(c, 0, sizeof *c); free(c); }
}
uint32_t afxo_ctx_n(const afxo_ctx* c) { return c->n; }

/* SystemParameters::hash_and_pray, src/parameters.rs:196-326.  `stream` stands in for csprng.fill_bytes
 * (32 bytes per attempt).  Returns bytes consumed, or -1 (stream exhausted) / -2 (NoSystemParameters). */
long afxo_system_parameters_generate(uint32_t n, const uint8_t* stream, size_t stream_len, uint8_t* out) {
  afxo_init_constants();
  if (n == 0 || n > AFX_MAX_ATTRIBUTES) return -2;
  uint32_t g = n < 3 ? 3 : n;
  size_t used = 0;
  uint32_t total = 4 + g + n + 4; /* G_w, G_w', G_x0, G_x1, G_y.., G_m.., G_V, G_a, G_a0, G_a1 */
  uint8_t enc[4 + 2 * AFX_MAX_ATTRIBUTES + 4 + 3][32];
  for (uint32_t k = 0; k < total; k++) {
    for (;;) {
      if (used + 32 > stream_len) return -1;
      ge tmp;
      int ok = ristretto_decode(&tmp, stream + used);
      memcpy(enc[k], stream + used, 32);
      used += 32;
      if (ok) break;
    }
  }
  /* uniqueness / non-identity check, src/parameters.rs:297-323 (G_y, G_m only up to n; the loop
   * there skips comparing against the last remaining element — restated literally) */
  uint8_t gens[10 + 2 * AFX_MAX_ATTRIBUTES][32];
  int ng = 0;
  ge B;
  ristretto_basepoint(&B);
  memset(gens[ng++], 0, 32);
  ristretto_encode(gens[ng++], &B);
  memcpy(gens[ng++], enc[0], 32);            /* G_w   */
  memcpy(gens[ng++], enc[1], 32);            /* G_w'  */
  memcpy(gens[ng++], enc[2], 32);            /* G_x0  */
  memcpy(gens[ng++], enc[3], 32);            /* G_x1  */
  memcpy(gens[ng++], enc[4 + g + n], 32);    /* G_V   */
  memcpy(gens[ng++], enc[4 + g + n + 1], 32);
  memcpy(gens[ng++], enc[4 + g + n + 2], 32);
  memcpy(gens[ng++], enc[4 + g + n + 3], 32);
  for (uint32_t i = 0; i < n; i++) {
    memcpy(gens[ng++], enc[4 + i], 32);
    memcpy(gens[ng++], enc[4 + g + i], 32);
  }
  while (ng >= 2) {
    ng--;
    for (int i = 0; i + 1 < ng; i++)
      if (memcmp(gens[ng], gens[i], 32) == 0) return -2;
  }
  /* to_bytes, src/parameters.rs:155-184 */
  uint8_t* p = out;
  wr32(p, n); p += 4;
  ristretto_encode(p, &B); p += 32;
  memcpy(p, enc[0], 32 * 4); p += 32 * 4;
  memcpy(p, enc[4], 32 * (size_t)g); p += 32 * (size_t)g;
  memcpy(p, enc[4 + g], 32 * (size_t)n); p += 32 * (size_t)n;
  memcpy(p, enc[4 + g + n], 32 * 4);
  return (long)used;
}

/* Issuer::new, src/issuer.rs:78-93: SecretKey::generate (src/amacs.rs:89-107; 4+n draws of 64 B) +
 * IssuerParameters::generate (src/parameters.rs:349-362) */
int afxo_issuer_new(const uint8_t* params, size_t plen, const uint8_t* draws, uint8_t* key_out, uint8_t issuer_params_out[64]) {
  afxo_ctx* c = afxo_ctx_new(params, plen, NULL, 0, NULL);
  if (!c) return -1;
  uint32_t n = c->n;
  sc w, wp, x0, x1, y[AFX_MAX_ATTRIBUTES];
  sc_reduce_wide(&w, draws);
  sc_reduce_wide(&wp, draws + 64);
  sc_reduce_wide(&x0, draws + 128);
  sc_reduce_wide(&x1, draws + 192);
  for (uint32_t i = 0; i < n; i++) sc_reduce_wide(&y[i], draws + 256 + 64 * (size_t)i);
  ge W, t, C_W, I;
  ge_scalarmult(&W, &w, &c->G_w);
  uint8_t* p = key_out;
  wr32(p, n); p += 4;
  memcpy(p, w.b, 32); p += 32; memcpy(p, wp.b, 32); p += 32; memcpy(p, x0.b, 32); p += 32; memcpy(p, x1.b, 32); p += 32;
  for (uint32_t i = 0; i < n; i++) { memcpy(p, y[i].b, 32); p += 32; }
  ristretto_encode(p, &W);
  ge_scalarmult(&C_W, &w, &c->G_w);
  ge_scalarmult(&t, &wp, &c->G_wp);
  ge_add(&C_W, &C_W, &t);
  I = c->G_V;
  ge_scalarmult(&t, &x0, &c->G_x0); ge_sub(&I, &I, &t);
  ge_scalarmult(&t, &x1, &c->G_x1); ge_sub(&I, &I, &t);
  for (uint32_t i = 0; i < n; i++) { ge_scalarmult(&t, &y[i], &c->G_y[i]); ge_sub(&I, &I, &t); }
  ristretto_encode(issuer_params_out, &C_W);
  ristretto_encode(issuer_params_out + 32, &I);
  afxo_ctx_free(c);
  return 0;
}

/* ---- symmetric.rs / encoding.rs ---- */

/* encode_to_group, src/encoding.rs:56-70.  Returns counter, or -1 for the reference's panic. */
static int encode_to_group(ge* out, const uint8_t* data, size_t len) {
  uint8_t bytes[32];
  memset(bytes, 0, 32);
  memcpy(bytes + 1, data, len);
  for (int j = 0; j < 64; j++) {
    bytes[31] = (uint8_t)j;
    for (int i = 0; i < 128; i++) {
      bytes[0] = (uint8_t)(2 * i);
      if (ristretto_decode(out, bytes)) return i + j * 128;
    }
  }
  return -1;
}

int afxo_encode_to_group(const uint8_t* data, size_t len, uint8_t out[32]) {
  afxo_init_constants();
  if (len > 30) return -1;
  ge p;
  int ctr = encode_to_group(&p, data, len);
  if (ctr >= 0) ristretto_encode(out, &p);
  return ctr;
}

/* decode_from_group, src/encoding.rs:75-82 */
int afxo_decode_from_group(const uint8_t pt[32], uint8_t data[30]) {
  memcpy(data, pt + 1, 30);
  return (pt[0] / 2) + pt[31] * 128;
}

static void hash_to_point(ge* out, const uint8_t* msg, size_t len) {
  uint8_t h[64];
  afxo_sha512(h, msg, len);
  ristretto_from_uniform_bytes(out, h);
}
static void hash_to_scalar(sc* out, const uint8_t* msg, size_t len) {
  uint8_t h[64];
  afxo_sha512(h, msg, len);
  sc_reduce_wide(out, h);
}

/* impl From<&[u8; 30]> for Plaintext, src/symmetric.rs:135-143.  out = M1 || M2 || m3. */
int afxo_plaintext_from_bytes(const uint8_t msg[30], uint8_t out[96]) {
  afxo_init_constants();
  ge M1, M2;
  sc m3;
  int ctr = encode_to_group(&M1, msg, 30);
  if (ctr < 0) return -1;
  hash_to_point(&M2, msg, 30);
  hash_to_scalar(&m3, msg, 30);
  ristretto_encode(out, &M1);
  ristretto_encode(out + 32, &M2);
  memcpy(out + 64, m3.b, 32);
  return ctr;
}

/* Keypair::derive, src/symmetric.rs:197-215.  out = a || a0 || a1 || pk */
int afxo_keypair_derive(const afxo_ctx* c, const uint8_t master_secret[64], uint8_t out[128]) {
  sc a, a0, a1;
  hash_to_scalar(&a, master_secret, 64);
  hash_to_scalar(&a0, a.b, 32);
  hash_to_scalar(&a1, a0.b, 32);
  ge pk, t;
  ge_scalarmult(&pk, &a, &c->G_a);
  ge_scalarmult(&t, &a0, &c->G_a0); ge_add(&pk, &pk, &t);
  ge_scalarmult(&t, &a1, &c->G_a1); ge_add(&pk, &pk, &t);
  memcpy(out, a.b, 32); memcpy(out + 32, a0.b, 32); memcpy(out + 64, a1.b, 32);
  ristretto_encode(out + 96, &pk);
  return 0;
}

/* Keypair::encrypt, src/symmetric.rs:252-261 */
static void sym_encrypt(ge* E1, ge* E2, const sc* a, const sc* a0, const sc* a1, const ge* M1, const ge* M2, const sc* m3) {
  sc k;
  sc_muladd(&k, a1, m3, a0);
  ge_scalarmult(E1, &k, M2);
  ge_scalarmult(E2, a, E1);
  ge_add(E2, E2, M1);
}

int afxo_encrypt(const uint8_t keypair[128], const uint8_t plaintext[96], uint8_t out[64]) {
  afxo_init_constants();
  sc a, a0, a1, m3;
  ge M1, M2, E1, E2;
  memcpy(a.b, keypair, 32); memcpy(a0.b, keypair + 32, 32); memcpy(a1.b, keypair + 64, 32);
  if (!ristretto_decode(&M1, plaintext) || !ristretto_decode(&M2, plaintext + 32)) return -1;
  memcpy(m3.b, plaintext + 64, 32);
  sym_encrypt(&E1, &E2, &a, &a0, &a1, &M1, &M2, &m3);
  ristretto_encode(out, &E1);
  ristretto_encode(out + 32, &E2);
  return 0;
}

/* Keypair::decrypt, src/symmetric.rs:273-289.  0 ok; 1 UndecryptableAttribute */
int afxo_decrypt(const uint8_t keypair[128], const uint8_t ciphertext[64], uint8_t plaintext_out[96]) {
  afxo_init_constants();
  sc a, a0, a1, m3p, k;
  ge E1, E2, M1p, M2p, E1p, t;
  memcpy(a.b, keypair, 32); memcpy(a0.b, keypair + 32, 32); memcpy(a1.b, keypair + 64, 32);
  if (!ristretto_decode(&E1, ciphertext) || !ristretto_decode(&E2, ciphertext + 32)) return -1;
  ge_scalarmult(&t, &a, &E1);
  ge_sub(&M1p, &E2, &t);
  uint8_t enc[32], m[30];
  ristretto_encode(enc, &M1p);
  afxo_decode_from_group(enc, m);
  hash_to_scalar(&m3p, m, 30);
  hash_to_point(&M2p, m, 30);
  sc_muladd(&k, &a1, &m3p, &a0);
  ge_scalarmult(&E1p, &k, &M2p);
  if (!ristretto_eq(&E1, &E1p)) return 1;
  memcpy(plaintext_out, enc, 32);
  ristretto_encode(plaintext_out + 32, &M2p);
  memcpy(plaintext_out + 64, m3p.b, 32);
  return 0;
}

/* ---- attributes ---- */
typedef struct {
  int kind;
  sc s;        /* scalar kinds */
  ge M1, M2;   /* point kinds: M1 = the point; M2/m3 only for plaintext kinds */
  sc m3;
} attr_t;

static int parse_attrs(attr_t* a, uint32_t n, const uint8_t* kinds, const uint8_t* values /* [n][96] */) {
  for (uint32_t i = 0; i < n; i++) {
    const uint8_t* v = values + 96 * (size_t)i;
    a[i].kind = kinds[i];
    switch (kinds[i]) {
      case AFX_ATTR_PUBLIC_SCALAR: case AFX_ATTR_SECRET_SCALAR:
        if (!sc_is_canonical(v)) return 0;
        memcpy(a[i].s.b, v, 32);
        break;
      case AFX_ATTR_PUBLIC_POINT:
        if (!ristretto_decode(&a[i].M1, v)) return 0;
        break;
      case AFX_ATTR_EITHER_POINT: case AFX_ATTR_SECRET_POINT:
        if (!ristretto_decode(&a[i].M1, v) || !ristretto_decode(&a[i].M2, v + 32)) return 0;
        if (!sc_is_canonical(v + 64)) return 0;
        memcpy(a[i].m3.b, v + 64, 32);
        break;
      default: return 0;
    }
  }
  return 1;
}

/* Messages::from_attributes, src/amacs.rs:225-243 */
static void messages_from_attributes(ge* M, const afxo_ctx* c, const attr_t* a, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    if (a[i].kind == AFX_ATTR_PUBLIC_SCALAR || a[i].kind == AFX_ATTR_SECRET_SCALAR) ge_scalarmult(&M[i], &a[i].s, &c->G_m[i]);
    else M[i] = a[i].M1;
  }
}

/* ---- ProofOfIssuance, src/nizk/issuance.rs ---- */
static void issuance_statement(zkp_cs* z, const afxo_ctx* c, int prover, const sc* t, const ge* U, const ge* V, const ge* M, uint32_t n_msgs) {
  uint32_t n = c->n, g = c->g;
  int w, wp, x0, x1, y[AFX_MAX_ATTRIBUTES], one;
  sc s_one;
  sc_one(&s_one);
  zkp_init(z, prover, "2019/1416 anonymous credential", "2019/1416 issuance proof");
  w = zkp_alloc_scalar(z, "w", prover ? &c->w : NULL);
  wp = zkp_alloc_scalar(z, "w'", prover ? &c->wp : NULL);
  x0 = zkp_alloc_scalar(z, "x_0", prover ? &c->x0 : NULL);
  x1 = zkp_alloc_scalar(z, "x_1", prover ? &c->x1 : NULL);
  for (uint32_t i = 0; i < n; i++) y[i] = zkp_alloc_scalar(z, "y", prover ? &c->y[i] : NULL);
  one = zkp_alloc_scalar(z, "1", prover ? &s_one : NULL);

  ge neg, tU;
  uint8_t enc[32];
#define PT(label, P) (prover ? zkp_alloc_point_prover(z, label, (P)) : (ristretto_encode(enc, (P)), zkp_alloc_point_verifier(z, label, enc)))
  int G_V = PT("G_V", &c->G_V);
  int G_w = PT("G_w", &c->G_w);
  int G_wp = PT("G_w_prime", &c->G_wp);
  ge_neg(&neg, &c->G_x0); int nGx0 = PT("-G_x_0", &neg);
  ge_neg(&neg, &c->G_x1); int nGx1 = PT("-G_x_1", &neg);
  int nGy[AFX_MAX_ATTRIBUTES];
  for (uint32_t i = 0; i < g; i++) { ge_neg(&neg, &c->G_y[i]); nGy[i] = PT("-G_y", &neg); }
  int C_W = PT("C_W", &c->C_W);
  int I = PT("I", &c->I);
  int Uv = PT("U", U);
  int Vv = PT("V", V);
  ge_scalarmult(&tU, t, U);
  int tUv = PT("tU", &tU);
  int Mv[AFX_MAX_ATTRIBUTES];
  for (uint32_t i = 0; i < n_msgs; i++) Mv[i] = PT("M", &M[i]);
#undef PT
  int scs[ZKP_MAX_TERMS], pts[ZKP_MAX_TERMS];
  scs[0] = w; pts[0] = G_w; scs[1] = wp; pts[1] = G_wp;
  zkp_constrain(z, C_W, 2, scs, pts);
  /* rhs.extend(y.zip(neg_G_y)): zip truncates to min(n, g) = n */
  int k = 0;
  scs[k] = one; pts[k++] = G_V; scs[k] = x0; pts[k++] = nGx0; scs[k] = x1; pts[k++] = nGx1;
  for (uint32_t i = 0; i < n; i++) { scs[k] = y[i]; pts[k++] = nGy[i]; }
  zkp_constrain(z, I, k, scs, pts);
  k = 0;
  scs[k] = w; pts[k++] = G_w; scs[k] = x0; pts[k++] = Uv; scs[k] = x1; pts[k++] = tUv;
  uint32_t nm = n_msgs < n ? n_msgs : n; /* y.zip(M) truncates */
  for (uint32_t i = 0; i < nm; i++) { scs[k] = y[i]; pts[k++] = Mv[i]; }
  zkp_constrain(z, Vv, k, scs, pts);
}

/* Issuer::issue, src/issuer.rs:111-124.  values: [n_attrs][96].  out_responses: (n+5)*32 */
int afxo_issue(const afxo_ctx* c, uint32_t n_attrs, const uint8_t* kinds, const uint8_t* values, const uint8_t t_wide[64],
               const uint8_t U_wide[64], const uint8_t rng_seed[32], uint8_t out_t[32], uint8_t out_U[32], uint8_t out_V[32],
               uint8_t out_challenge[32], uint8_t* out_responses) {
  if (!c->has_key || !c->has_issuer_params) return -1;
  /* Amac::tag, src/amacs.rs:285-287 */
  if (n_attrs != c->n) return AFX_ST_MAC_CREATION;
  attr_t a[AFX_MAX_ATTRIBUTES];
  if (!parse_attrs(a, n_attrs, kinds, values)) return -1;
  sc t, x1t;
  ge U, V, tmp, M[AFX_MAX_ATTRIBUTES];
  sc_reduce_wide(&t, t_wide);
  ristretto_from_uniform_bytes(&U, U_wide);
  /* compute_V, src/amacs.rs:256-272 */
  messages_from_attributes(M, c, a, n_attrs);
  V = c->W;
  ge_scalarmult(&tmp, &c->x0, &U); ge_add(&V, &V, &tmp);
  sc_mul(&x1t, &c->x1, &t);
  ge_scalarmult(&tmp, &x1t, &U); ge_add(&V, &V, &tmp);
  ge_multiscalar(&tmp, c->y, M, (int)n_attrs);
  ge_add(&V, &V, &tmp);
  memcpy(out_t, t.b, 32);
  ristretto_encode(out_U, &U);
  ristretto_encode(out_V, &V);
  /* ProofOfIssuance::prove, src/nizk/issuance.rs:40-129 (recomputes Messages at :95) */
  zkp_cs* z = (zkp_cs*)malloc(sizeof *z);
  issuance_statement(z, c, 1, &t, &U, &V, M, n_attrs);
  sc ch, resp[ZKP_MAX_SCALARS];
  zkp_prove_compact(z, rng_seed, &ch, resp);
  memcpy(out_challenge, ch.b, 32);
  for (uint32_t i = 0; i < c->n + 5; i++) memcpy(out_responses + 32 * (size_t)i, resp[i].b, 32);
  free(z);
  return AFX_ST_OK;
}

/* CredentialIssuance::verify, src/issuer.rs:48-57 -> ProofOfIssuance::verify, src/nizk/issuance.rs:132-218 */
int afxo_issuance_verify(const afxo_ctx* c, uint32_t n_attrs, const uint8_t* kinds, const uint8_t* values, const uint8_t t[32],
                         const uint8_t U[32], const uint8_t V[32], const uint8_t challenge[32], const uint8_t* responses,
                         uint32_t n_responses) {
  if (!c->has_issuer_params) return -1;
  attr_t a[AFX_MAX_ATTRIBUTES];
  if (n_attrs > AFX_MAX_ATTRIBUTES) return AFX_ST_VERIFICATION_FAILURE;
  if (!parse_attrs(a, n_attrs, kinds, values)) return AFX_ST_VERIFICATION_FAILURE;
  /* Messages::from_attributes indexes G_m[i]: more attributes than n panics the reference */
  if (n_attrs > c->n) return AFX_ST_VERIFICATION_FAILURE;
  sc ts;
  ge Up, Vp, M[AFX_MAX_ATTRIBUTES];
  if (!sc_is_canonical(t) || !sc_is_canonical(challenge)) return AFX_ST_VERIFICATION_FAILURE;
  for (uint32_t i = 0; i < n_responses; i++)
    if (!sc_is_canonical(responses + 32 * (size_t)i)) return AFX_ST_VERIFICATION_FAILURE;
  memcpy(ts.b, t, 32);
  if (!ristretto_decode(&Up, U) || !ristretto_decode(&Vp, V)) return AFX_ST_VERIFICATION_FAILURE;
  messages_from_attributes(M, c, a, n_attrs);
  zkp_cs* z = (zkp_cs*)malloc(sizeof *z);
  issuance_statement(z, c, 0, &ts, &Up, &Vp, M, n_attrs);
  int ok = zkp_verify_compact(z, challenge, responses, (int)n_responses);
  free(z);
  return ok ? AFX_ST_OK : AFX_ST_VERIFICATION_FAILURE;
}

/* ---- ProofOfEncryption, src/nizk/encryption.rs ---- */
typedef struct { ge pk, E1, E2, C_y_1, C_y_2, C_y_3, C_y_2p; } encpts_t;

static int encryption_statement(zkp_cs* z, const afxo_ctx* c, int prover, uint16_t index, const encpts_t* p, const sc* wit /* a,a0,a1,m3,z,z1 */) {
  if (index >= c->n) return 0; /* G_m[index] would panic, src/nizk/encryption.rs:100,179 */
  zkp_init(z, prover, "2019/1416 anonymous credentials", "2019/1416 proof of encryption");
  int a = zkp_alloc_scalar(z, "a", prover ? &wit[0] : NULL);
  int a0 = zkp_alloc_scalar(z, "a0", prover ? &wit[1] : NULL);
  int a1 = zkp_alloc_scalar(z, "a1", prover ? &wit[2] : NULL);
  int m3 = zkp_alloc_scalar(z, "m3", prover ? &wit[3] : NULL);
  int zz = zkp_alloc_scalar(z, "z", prover ? &wit[4] : NULL);
  int z1 = zkp_alloc_scalar(z, "z1", prover ? &wit[5] : NULL);
  uint8_t enc[32];
  ge d;
#define PT(label, P) (prover ? zkp_alloc_point_prover(z, label, (P)) : (ristretto_encode(enc, (P)), zkp_alloc_point_verifier(z, label, enc)))
  int pk = PT("pk", &p->pk);
  int G_a = PT("G_a", &c->G_a);
  int G_a0 = PT("G_a_0", &c->G_a0);
  int G_a1 = PT("G_a_1", &c->G_a1);
  int G_y1 = PT("G_y_1", &c->G_y[0]);
  int G_y2 = PT("G_y_2", &c->G_y[1]);
  int G_y3 = PT("G_y_3", &c->G_y[2]);
  int G_m3 = PT("G_m_3", &c->G_m[index]);
  int C_y_2 = PT("C_y_2", &p->C_y_2);
  int C_y_3 = PT("C_y_3", &p->C_y_3);
  int C_y_2p = PT("C_y_2'", &p->C_y_2p);
  ge_sub(&d, &p->C_y_1, &p->E2);
  int C_y_1_minus_E2 = PT("C_y_1-E2", &d);
  int E1 = PT("E1", &p->E1);
  ge_neg(&d, &p->E1);
  int mE1 = PT("-E1", &d);
#undef PT
  int s[3], q[3];
  s[0] = a; q[0] = G_a; s[1] = a0; q[1] = G_a0; s[2] = a1; q[2] = G_a1;
  zkp_constrain(z, pk, 3, s, q);
  s[0] = zz; q[0] = G_y1; s[1] = a; q[1] = mE1;
  zkp_constrain(z, C_y_1_minus_E2, 2, s, q);
  s[0] = a1; q[0] = C_y_2;
  zkp_constrain(z, C_y_2p, 1, s, q);
  s[0] = a0; q[0] = C_y_2; s[1] = m3; q[1] = C_y_2p; s[2] = z1; q[2] = G_y2;
  zkp_constrain(z, E1, 3, s, q);
  s[0] = zz; q[0] = G_y3; s[1] = m3; q[1] = G_m3;
  zkp_constrain(z, C_y_3, 2, s, q);
  return 1;
}

static int encproof_decode(encpts_t* p, const afxo_encproof* e) {
  return ristretto_decode(&p->pk, e->pk) && ristretto_decode(&p->E1, e->E1) && ristretto_decode(&p->E2, e->E2) &&
         ristretto_decode(&p->C_y_1, e->C_y_1) && ristretto_decode(&p->C_y_2, e->C_y_2) &&
         ristretto_decode(&p->C_y_3, e->C_y_3) && ristretto_decode(&p->C_y_2p, e->C_y_2p);
}

/* ProofOfEncryption::verify, src/nizk/encryption.rs:154-210 */
int afxo_verify_encryption_proof(const afxo_ctx* c, const afxo_encproof* e) {
  encpts_t p;
  if (!sc_is_canonical(e->challenge)) return AFX_ST_VERIFICATION_FAILURE;
  for (int i = 0; i < 6; i++)
    if (!sc_is_canonical(e->responses[i])) return AFX_ST_VERIFICATION_FAILURE;
  if (!encproof_decode(&p, e)) return AFX_ST_VERIFICATION_FAILURE;
  zkp_cs* z = (zkp_cs*)malloc(sizeof *z);
  int ok = encryption_statement(z, c, 0, e->index, &p, NULL);
  if (ok) ok = zkp_verify_compact(z, e->challenge, &e->responses[0][0], 6);
  free(z);
  return ok ? AFX_ST_OK : AFX_ST_VERIFICATION_FAILURE;
}

/* ProofOfEncryption::prove, src/nizk/encryption.rs:58-142 */
static int encryption_prove(afxo_encproof* out, const afxo_ctx* c, const attr_t* pt, uint16_t index, const sc* a, const sc* a0,
                            const sc* a1, const ge* pk, const sc* zn, const uint8_t seed[32]) {
  encpts_t p;
  ge t;
  sc wit[6], k;
  if (index >= c->n) return 0;
  sym_encrypt(&p.E1, &p.E2, a, a0, a1, &pt->M1, &pt->M2, &pt->m3);
  ge_scalarmult(&p.C_y_1, zn, &c->G_y[0]); ge_add(&p.C_y_1, &p.C_y_1, &pt->M1);
  ge_scalarmult(&p.C_y_2, zn, &c->G_y[1]); ge_add(&p.C_y_2, &p.C_y_2, &pt->M2);
  ge_scalarmult(&p.C_y_3, zn, &c->G_y[2]); ge_scalarmult(&t, &pt->m3, &c->G_m[index]); ge_add(&p.C_y_3, &p.C_y_3, &t);
  ge_scalarmult(&p.C_y_2p, a1, &p.C_y_2);
  /* z1 = -z (a0 + a1 m3) */
  sc_muladd(&k, a1, &pt->m3, a0);
  sc_mul(&k, zn, &k);
  sc_neg(&wit[5], &k);
  wit[0] = *a; wit[1] = *a0; wit[2] = *a1; wit[3] = pt->m3; wit[4] = *zn;
  p.pk = *pk;
  zkp_cs* z = (zkp_cs*)malloc(sizeof *z);
  encryption_statement(z, c, 1, index, &p, wit);
  sc ch, resp[ZKP_MAX_SCALARS];
  zkp_prove_compact(z, seed, &ch, resp);
  free(z);
  memcpy(out->challenge, ch.b, 32);
  for (int i = 0; i < 6; i++) memcpy(out->responses[i], resp[i].b, 32);
  ristretto_encode(out->pk, &p.pk);
  ristretto_encode(out->E1, &p.E1);
  ristretto_encode(out->E2, &p.E2);
  ristretto_encode(out->C_y_1, &p.C_y_1);
  ristretto_encode(out->C_y_2, &p.C_y_2);
  ristretto_encode(out->C_y_3, &p.C_y_3);
  ristretto_encode(out->C_y_2p, &p.C_y_2p);
  out->index = index;
  return 1;
}

/* ---- ProofOfValidCredential, src/nizk/presentation.rs ---- */

/* the transcript + constraint part shared by prove (:187-273) and verify (:355-433).
 * kinds are EncryptedAttribute kinds.  Returns 0 where the reference would panic. */
static int presentation_statement(zkp_cs* z, const afxo_ctx* c, int prover, uint32_t n_attrs, const uint8_t* kinds,
                                  uint32_t hs, const uint16_t* hidx, const sc* wit /* z, z_0, t, m... */, const ge* C_x_1,
                                  const ge* C_x_0, const ge* C_y /* [n_attrs] */, const ge* Z,
                                  const ge* C_y_1 /* strict mode: per hidden group element, in position order, the C_y_1 of its proof of encryption */) {
  zkp_init(z, prover, "2019/1416 anonymous credential", "2019/1416 presentation proof");
  int zz = zkp_alloc_scalar(z, "z", prover ? &wit[0] : NULL);
  int z0 = zkp_alloc_scalar(z, "z_0", prover ? &wit[1] : NULL);
  int t = zkp_alloc_scalar(z, "t", prover ? &wit[2] : NULL);
  int Hs[AFX_MAX_ATTRIBUTES];
  for (uint32_t k = 0; k < hs; k++) Hs[k] = zkp_alloc_scalar(z, "m", prover ? &wit[3 + k] : NULL);
  uint8_t enc[32];
#define PT(label, P) (prover ? zkp_alloc_point_prover(z, label, (P)) : (ristretto_encode(enc, (P)), zkp_alloc_point_verifier(z, label, enc)))
  int I = PT("I", &c->I);
  int Cx1 = PT("C_x_1", C_x_1);
  int Cx0 = PT("C_x_0", C_x_0);
  int Gx0 = PT("G_x_0", &c->G_x0);
  int Gx1 = PT("G_x_1", &c->G_x1);
  int Cy[AFX_MAX_ATTRIBUTES], Gy[AFX_MAX_ATTRIBUTES], Gm[AFX_MAX_ATTRIBUTES];
  uint32_t k = 0;
  for (uint32_t i = 0; i < n_attrs; i++) {
    if (kinds[i] == AFX_ENC_SECRET_POINT) continue;
    Cy[k++] = PT("C_y", &C_y[i]);
  }
  for (uint32_t i = 0; i < c->g; i++) Gy[i] = PT("G_y", &c->G_y[i]);
  for (uint32_t j = 0; j < hs; j++) {
    if (hidx[j] >= c->n) return 0; /* G_m[*i] out of range: panic at presentation.rs:407 */
    Gm[j] = PT("G_m", &c->G_m[hidx[j]]);
  }
  /* strict mode, the DLEQ the reference's README.md:121-122 lists as TODO: the plaintext committed to in C_y[i] is the one the
   * proof of encryption is about.  C_y[i] = z G_y[i] + M1 (presentation.rs:173) and C_y_1 = z G_y[0] + M1 (encryption.rs:70), so
   * C_y[i] - C_y_1 = z G_y[i] + z (-G_y[0]) with the presentation's own z.  For a hidden group element at position 0 the
   * difference must be the identity, which is checked directly (an identity point cannot enter a zkp transcript). */
  int Dv[AFX_MAX_ATTRIBUTES], Dpos[AFX_MAX_ATTRIBUTES], nD = 0, negGy1 = -1;
  if (c->strict && C_y_1) {
    uint32_t e = 0;
    for (uint32_t i = 0; i < n_attrs; i++) {
      if (kinds[i] != AFX_ENC_SECRET_POINT) continue;
      ge D;
      ge_sub(&D, &C_y[i], &C_y_1[e]);
      e++;
      if (i == 0) {
        uint8_t de[32];
        static const uint8_t zero[32] = { 0 };
        ristretto_encode(de, &D);
        if (memcmp(de, zero, 32) != 0) return 0;
        continue;
      }
      if (negGy1 < 0) {
        ge ng;
        ge_neg(&ng, &c->G_y[0]);
        negGy1 = PT("-G_y_1", &ng);
      }
      Dv[nD] = PT("C_y-C_y_1", &D);
      Dpos[nD++] = (int)i;
    }
  }
  int Zv = PT("Z", Z);
#undef PT
  int s[3], q[3];
  s[0] = zz; q[0] = I;
  zkp_constrain(z, Zv, 1, s, q);
  s[0] = t; q[0] = Cx0; s[1] = z0; q[1] = Gx0; s[2] = zz; q[2] = Gx1;
  zkp_constrain(z, Cx1, 3, s, q);
  if (c->strict) {
    /* strict mode: the statement the scheme intends - the j-th kept commitment is checked against the generators and
     * the kind of ITS OWN position */
    uint32_t j = 0;
    for (uint32_t i = 0; i < n_attrs; i++) {
      if (kinds[i] == AFX_ENC_SECRET_POINT) continue;
      if (kinds[i] == AFX_ENC_SECRET_SCALAR) {
        int found = -1;
        for (uint32_t h = 0; h < hs; h++)
          if (hidx[h] == i) { found = (int)h; break; }
        if (found < 0) return 0;
        s[0] = zz; q[0] = Gy[i]; s[1] = Hs[found]; q[1] = Gm[found];
        zkp_constrain(z, Cy[j], 2, s, q);
      } else {
        s[0] = zz; q[0] = Gy[i];
        zkp_constrain(z, Cy[j], 1, s, q);
      }
      j++;
    }
    for (int d = 0; d < nD; d++) {
      s[0] = zz; q[0] = Gy[Dpos[d]]; s[1] = zz; q[1] = negGy1;
      zkp_constrain(z, Dv[d], 2, s, q);
    }
    return 1;
  }
  /* constraint #3, restated literally (compact index used as original position) */
  for (uint32_t j = 0; j < k; j++) {
    if (j >= n_attrs) return 0;       /* encrypted_attributes[i] out of range */
    if (kinds[j] == AFX_ENC_SECRET_POINT) continue;
    if (j >= c->g) return 0;          /* G_y[i] out of range */
    if (kinds[j] == AFX_ENC_SECRET_SCALAR) {
      /* H_s[i], G_m[i]: association-list lookup by original index; absent => panic (:81,:100) */
      int found = -1;
      for (uint32_t h = 0; h < hs; h++)
        if (hidx[h] == j) { found = (int)h; break; }
      if (found < 0) return 0;
      s[0] = zz; q[0] = Gy[j]; s[1] = Hs[found]; q[1] = Gm[found];
      zkp_constrain(z, Cy[j], 2, s, q);
    } else {
      s[0] = zz; q[0] = Gy[j];
      zkp_constrain(z, Cy[j], 1, s, q);
    }
  }
  return 1;
}

/* AnonymousCredential::show, src/credential.rs:37-46 -> ProofOfValidCredential::prove, presentation.rs:139-321.
 * values: [n][96]; keypair: a||a0||a1||pk or NULL; enc_seeds: [#SecretPoint][32] */
int afxo_show(const afxo_ctx* c, uint32_t n_attrs, const uint8_t* kinds, const uint8_t* values, const uint8_t t_in[32],
              const uint8_t U_in[32], const uint8_t V_in[32], const uint8_t* keypair, const uint8_t z_wide[64],
              const uint8_t rng_seed[32], const uint8_t* enc_seeds, afxo_presentation* out) {
  if (!c->has_issuer_params) return -1;
  if (n_attrs > AFX_MAX_ATTRIBUTES || n_attrs > c->g) return -1; /* G_y[i] index would panic */
  attr_t a[AFX_MAX_ATTRIBUTES];
  if (!parse_attrs(a, n_attrs, kinds, values)) return -1;
  if (!keypair)
    for (uint32_t i = 0; i < n_attrs; i++)
      if (kinds[i] == AFX_ATTR_SECRET_POINT) return AFX_ST_NO_SYMMETRIC_KEY;
  sc t, zn, z0, wit[3 + AFX_MAX_ATTRIBUTES];
  ge U, V, tmp;
  memcpy(t.b, t_in, 32);
  if (!ristretto_decode(&U, U_in) || !ristretto_decode(&V, V_in)) return -1;
  sc_reduce_wide(&zn, z_wide);
  sc_mul(&z0, &t, &zn);
  sc_neg(&z0, &z0);
  ge C_y[AFX_MAX_ATTRIBUTES], C_x_0, C_x_1, C_V, Z;
  uint32_t hs = 0;
  uint16_t hidx[AFX_MAX_ATTRIBUTES];
  uint8_t ekinds[AFX_MAX_ATTRIBUTES];
  wit[0] = zn; wit[1] = z0; wit[2] = t;
  for (uint32_t i = 0; i < n_attrs; i++) {
    ge_scalarmult(&C_y[i], &zn, &c->G_y[i]);
    switch (a[i].kind) {
      case AFX_ATTR_SECRET_POINT: ge_add(&C_y[i], &C_y[i], &a[i].M1); ekinds[i] = AFX_ENC_SECRET_POINT; break;
      case AFX_ATTR_SECRET_SCALAR:
        if (i >= c->n) return -1;
        ge_scalarmult(&tmp, &a[i].s, &c->G_m[i]);
        ge_add(&C_y[i], &C_y[i], &tmp);
        wit[3 + hs] = a[i].s;
        hidx[hs++] = (uint16_t)i;
        ekinds[i] = AFX_ENC_SECRET_SCALAR;
        break;
      case AFX_ATTR_PUBLIC_SCALAR: ekinds[i] = AFX_ENC_PUBLIC_SCALAR; break;
      default: ekinds[i] = AFX_ENC_PUBLIC_POINT; break;
    }
  }
  ge_scalarmult(&C_x_0, &zn, &c->G_x0); ge_add(&C_x_0, &C_x_0, &U);
  ge_scalarmult(&C_x_1, &zn, &c->G_x1); ge_scalarmult(&tmp, &t, &U); ge_add(&C_x_1, &C_x_1, &tmp);
  ge_scalarmult(&C_V, &zn, &c->G_V); ge_add(&C_V, &C_V, &V);
  ge_scalarmult(&Z, &zn, &c->I);
  ge C_y_1[AFX_MAX_ATTRIBUTES];
  uint32_t nsp = 0;
  for (uint32_t i = 0; i < n_attrs; i++)
    if (ekinds[i] == AFX_ENC_SECRET_POINT) {   /* C_y_1 of the proof of encryption made below (encryption.rs:70) */
      ge_scalarmult(&C_y_1[nsp], &zn, &c->G_y[0]);
      ge_add(&C_y_1[nsp], &C_y_1[nsp], &a[i].M1);
      nsp++;
    }
  zkp_cs* z = (zkp_cs*)malloc(sizeof *z);
  if (!presentation_statement(z, c, 1, n_attrs, ekinds, hs, hidx, wit, &C_x_1, &C_x_0, C_y, &Z, c->strict ? C_y_1 : NULL)) { free(z); return -1; }
  sc ch, resp[ZKP_MAX_SCALARS];
  zkp_prove_compact(z, rng_seed, &ch, resp);
  free(z);
  memset(out, 0, sizeof *out);
  out->n_attributes = n_attrs;
  out->n_responses = 3 + hs;
  memcpy(out->challenge, ch.b, 32);
  for (uint32_t i = 0; i < 3 + hs; i++) memcpy(out->responses[i], resp[i].b, 32);
  ristretto_encode(out->C_x_0, &C_x_0);
  ristretto_encode(out->C_x_1, &C_x_1);
  ristretto_encode(out->C_V, &C_V);
  out->n_hidden_scalars = hs;
  memcpy(out->hidden_scalar_indices, hidx, sizeof(uint16_t) * hs);
  uint32_t ne = 0;
  for (uint32_t i = 0; i < n_attrs; i++) {
    ristretto_encode(out->C_y[i], &C_y[i]);
    out->kinds[i] = ekinds[i];
    if (ekinds[i] == AFX_ENC_PUBLIC_SCALAR) memcpy(out->attr_values[i], a[i].s.b, 32);
    else if (ekinds[i] == AFX_ENC_PUBLIC_POINT) ristretto_encode(out->attr_values[i], &a[i].M1);
    else if (ekinds[i] == AFX_ENC_SECRET_POINT) {
      sc ka, ka0, ka1;
      ge pk;
      memcpy(ka.b, keypair, 32); memcpy(ka0.b, keypair + 32, 32); memcpy(ka1.b, keypair + 64, 32);
      if (!ristretto_decode(&pk, keypair + 96)) return -1;
      if (!encryption_prove(&out->enc[ne], c, &a[i], (uint16_t)i, &ka, &ka0, &ka1, &pk, &zn, enc_seeds + 32 * (size_t)ne)) return -1;
      ne++;
    }
  }
  out->n_enc_proofs = ne;
  return AFX_ST_OK;
}

/* Issuer::verify, src/issuer.rs:141-147 -> ProofOfValidCredential::verify, presentation.rs:324-443 */
/* Strict mode (opt-in, off by default, NOT bit-compatible with the reference):
 *  - constraint #3 of the presentation proof uses each kept commitment's own position (prover and verifier), so
 *    presentations with hidden group elements anywhere verify (the reference only handles trailing ones, App. B);
 *  - the verifier requires exactly one proof of encryption per hidden group element, in position order
 *    (the reference verifies whatever is attached, presentation.rs:438-440).
 *  - the presentation proof also shows, with its own z, that C_y[i] - C_y_1 = z (G_y[i] - G_y[0]) for every hidden group
 *    element i and the C_y_1 of its proof of encryption: the DLEQ the reference's README.md:121-122 lists as TODO. */
void afxo_ctx_set_strict(afxo_ctx* c, int strict) { c->strict = strict != 0; }

int afxo_verify_presentation(const afxo_ctx* c, const afxo_presentation* p) {
  if (!c->has_key || !c->has_issuer_params) return -1;
  uint32_t n_attrs = p->n_attributes;
  if (c->strict && n_attrs <= AFX_MAX_ATTRIBUTES && p->n_enc_proofs <= AFX_MAX_ATTRIBUTES) {
    uint32_t e = 0;
    for (uint32_t i = 0; i < n_attrs; i++)
      if (p->kinds[i] == AFX_ENC_SECRET_POINT) {
        if (e >= p->n_enc_proofs || p->enc[e].index != i) return AFX_ST_VERIFICATION_FAILURE;
        e++;
      }
    if (e != p->n_enc_proofs) return AFX_ST_VERIFICATION_FAILURE;
  }
  if (n_attrs > AFX_MAX_ATTRIBUTES || n_attrs > c->n) return AFX_ST_VERIFICATION_FAILURE; /* y[i]/G_m[i] index panic */
  if (p->n_responses > 3 + AFX_MAX_ATTRIBUTES || p->n_hidden_scalars > AFX_MAX_ATTRIBUTES || p->n_enc_proofs > AFX_MAX_ATTRIBUTES)
    return AFX_ST_VERIFICATION_FAILURE;
  if (!sc_is_canonical(p->challenge)) return AFX_ST_VERIFICATION_FAILURE;
  for (uint32_t i = 0; i < p->n_responses; i++)
    if (!sc_is_canonical(p->responses[i])) return AFX_ST_VERIFICATION_FAILURE;
  ge C_x_0, C_x_1, C_V, C_y[AFX_MAX_ATTRIBUTES], Z, x, t;
  if (!ristretto_decode(&C_x_0, p->C_x_0) || !ristretto_decode(&C_x_1, p->C_x_1) || !ristretto_decode(&C_V, p->C_V))
    return AFX_ST_VERIFICATION_FAILURE;
  for (uint32_t i = 0; i < n_attrs; i++)
    if (!ristretto_decode(&C_y[i], p->C_y[i])) return AFX_ST_VERIFICATION_FAILURE;
  /* :342-352 — separate `point * scalar` products, as the reference writes them */
  ge_sub(&Z, &C_V, &c->W);
  ge_scalarmult(&t, &c->x0, &C_x_0); ge_sub(&Z, &Z, &t);
  ge_scalarmult(&t, &c->x1, &C_x_1); ge_sub(&Z, &Z, &t);
  for (uint32_t i = 0; i < n_attrs; i++) {
    switch (p->kinds[i]) {
      case AFX_ENC_PUBLIC_SCALAR: {
        sc m;
        if (!sc_is_canonical(p->attr_values[i])) return AFX_ST_VERIFICATION_FAILURE;
        memcpy(m.b, p->attr_values[i], 32);
        ge_scalarmult(&t, &m, &c->G_m[i]);
        ge_add(&x, &C_y[i], &t);
        break;
      }
      case AFX_ENC_PUBLIC_POINT: {
        ge M;
        if (!ristretto_decode(&M, p->attr_values[i])) return AFX_ST_VERIFICATION_FAILURE;
        ge_add(&x, &C_y[i], &M);
        break;
      }
      case AFX_ENC_SECRET_SCALAR: case AFX_ENC_SECRET_POINT: x = C_y[i]; break;
      default: return AFX_ST_VERIFICATION_FAILURE;
    }
    ge_scalarmult(&t, &c->y[i], &x);
    ge_sub(&Z, &Z, &t);
  }
  ge C_y_1[AFX_MAX_ATTRIBUTES];
  if (c->strict)
    for (uint32_t e = 0; e < p->n_enc_proofs; e++)
      if (!ristretto_decode(&C_y_1[e], p->enc[e].C_y_1)) return AFX_ST_VERIFICATION_FAILURE;
  zkp_cs* z = (zkp_cs*)malloc(sizeof *z);
  int ok = presentation_statement(z, c, 0, n_attrs, p->kinds, p->n_hidden_scalars, p->hidden_scalar_indices, NULL, &C_x_1, &C_x_0, C_y, &Z,
                                  c->strict ? C_y_1 : NULL);
  if (ok) ok = zkp_verify_compact(z, p->challenge, &p->responses[0][0], (int)p->n_responses);
  free(z);
  if (!ok) return AFX_ST_VERIFICATION_FAILURE;
  /* :438-440 */
  for (uint32_t e = 0; e < p->n_enc_proofs; e++)
    if (afxo_verify_encryption_proof(c, &p->enc[e]) != AFX_ST_OK) return AFX_ST_VERIFICATION_FAILURE;
  return AFX_ST_OK;
}

/* ---- primitive wrappers for KAT tests ---- */
int afxo_point_decode_encode(const uint8_t in[32], uint8_t out[32]) {
  afxo_init_constants();
  ge p;
  if (!ristretto_decode(&p, in)) return 0;
  ristretto_encode(out, &p);
  return 1;
}
void afxo_point_from_uniform(const uint8_t in[64], uint8_t out[32]) {
  afxo_init_constants();
  ge p;
  ristretto_from_uniform_bytes(&p, in);
  ristretto_encode(out, &p);
}
int afxo_point_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
  afxo_init_constants();
  ge p, q;
  if (!ristretto_decode(&p, a) || !ristretto_decode(&q, b)) return 0;
  ge_add(&p, &p, &q);
  ristretto_encode(out, &p);
  return 1;
}
int afxo_point_sub(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
  afxo_init_constants();
  ge p, q;
  if (!ristretto_decode(&p, a) || !ristretto_decode(&q, b)) return 0;
  ge_sub(&p, &p, &q);
  ristretto_encode(out, &p);
  return 1;
}
int afxo_point_scalarmult(const uint8_t s[32], const uint8_t a[32], uint8_t out[32]) {
  afxo_init_constants();
  ge p;
  sc k;
  if (!ristretto_decode(&p, a)) return 0;
  memcpy(k.b, s, 32);
  ge_scalarmult(&p, &k, &p);
  ristretto_encode(out, &p);
  return 1;
}
void afxo_basepoint(uint8_t out[32]) {
  afxo_init_constants();
  ge b;
  ristretto_basepoint(&b);
  ristretto_encode(out, &b);
}
/* out = sum s_k P_k ; vartime=1 uses the NAF path, 0 the radix-16 path (must agree) */
int afxo_multiscalar(uint32_t n, const uint8_t* scalars, const uint8_t* points, int vartime, uint8_t out[32]) {
  afxo_init_constants();
  if (n > ZKP_MAX_TERMS) return 0;
  sc s[ZKP_MAX_TERMS];
  ge p[ZKP_MAX_TERMS], r;
  for (uint32_t i = 0; i < n; i++) {
    memcpy(s[i].b, scalars + 32 * (size_t)i, 32);
    if (!ristretto_decode(&p[i], points + 32 * (size_t)i)) return 0;
  }
  if (vartime) ge_multiscalar_vartime(&r, s, p, (int)n);
  else ge_multiscalar(&r, s, p, (int)n);
  ristretto_encode(out, &r);
  return 1;
}
void afxo_scalar_reduce_wide(const uint8_t in[64], uint8_t out[32]) { sc r; sc_reduce_wide(&r, in); memcpy(out, r.b, 32); }
void afxo_scalar_muladd(const uint8_t a[32], const uint8_t b[32], const uint8_t c[32], uint8_t out[32]) {
  sc x, y, z, r;
  memcpy(x.b, a, 32); memcpy(y.b, b, 32); memcpy(z.b, c, 32);
  sc_muladd(&r, &x, &y, &z);
  memcpy(out, r.b, 32);
}
void afxo_scalar_neg(const uint8_t a[32], uint8_t out[32]) { sc x, r; memcpy(x.b, a, 32); sc_neg(&r, &x); memcpy(out, r.b, 32); }
int afxo_scalar_is_canonical(const uint8_t a[32]) { return sc_is_canonical(a); }
void afxo_keccak_f1600(uint8_t st[200]) { keccak_f1600(st); }
/* merlin KAT helper: Transcript::new(label); append_message(l1, m1); challenge_bytes(l2, out) */
void afxo_merlin_simple(const uint8_t* label, size_t llen, const uint8_t* l1, size_t l1len, const uint8_t* m1, size_t m1len,
                        const uint8_t* l2, size_t l2len, uint8_t* out, size_t outlen) {
  merlin_transcript t;
  merlin_new(&t, label, llen);
  merlin_append_message(&t, l1, l1len, m1, m1len);
  merlin_challenge_bytes(&t, l2, l2len, out, outlen);
}
/* A scripted merlin transcript (the script of include/aeonflux_gpu.h afx_merlin_challenges, for ONE item, with two liberties the
 * checker may take: any number of challenges, and operation 5 = append_message(label, the most recent challenge's bytes)): merlin's
 * published conformance vectors (merlin tests::equivalence_simple / equivalence_complex [3P]) run through the oracle's own
 * strobe / merlin layer with it.  fields: [n_fields][32].  Every challenge's bytes are appended to out; returns the bytes written,
 * -1 for a malformed script or a full buffer. */
long afxo_merlin_script(const uint8_t* script, size_t len, const uint8_t* fields, uint32_t n_fields, uint8_t* out, size_t cap) {
  size_t at = 0, wrote = 0;
  uint8_t last[64];
  uint32_t last_len = 0;
  merlin_transcript t;
  int opened = 0;
  while (at < len) {
    const uint8_t op = script[at++];
    if (len - at < 4) return -1;
    uint32_t llen;
    memcpy(&llen, script + at, 4);
    at += 4;
    if (len - at < llen) return -1;
    const uint8_t* lab = script + at;
    at += llen;
    if (op == 1) {
      if (opened) return -1;
      merlin_new(&t, lab, llen);
      opened = 1;
      continue;
    }
    if (!opened) return -1;
    if (op == 2) {
      uint32_t mlen;
      if (len - at < 4) return -1;
      memcpy(&mlen, script + at, 4);
      at += 4;
      if (len - at < mlen) return -1;
      merlin_append_message(&t, lab, llen, script + at, mlen);
      at += mlen;
    } else if (op == 3 || op == 4) {
      uint32_t v;
      if (len - at < 4) return -1;
      memcpy(&v, script + at, 4);
      at += 4;
      if (op == 3) {
        if (v >= n_fields) return -1;
        merlin_append_message(&t, lab, llen, fields + 32 * (size_t)v, 32);
      } else {
        if (v == 0 || v > 64 || cap - wrote < v) return -1;
        merlin_challenge_bytes(&t, lab, llen, last, v);
        last_len = v;
        memcpy(out + wrote, last, v);
        wrote += v;
      }
    } else if (op == 5) {
      merlin_append_message(&t, lab, llen, last, last_len);
    } else return -1;
  }
  return (long)wrote;
}
void afxo_debug_reset(void) {
  zkp_debug_ncommit = 0;
  memset(zkp_debug_commit, 0, sizeof zkp_debug_commit);
  memset(zkp_debug_challenge, 0, sizeof zkp_debug_challenge);
}
void afxo_debug_last(uint8_t* commits, int* ncommit, uint8_t challenge[32]) {
  *ncommit = zkp_debug_ncommit;
  memcpy(commits, zkp_debug_commit, 32 * (size_t)zkp_debug_ncommit);
  memcpy(challenge, zkp_debug_challenge, 32);
}
