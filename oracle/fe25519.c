/* ORACLE — test infrastructure only (see afx_oracle_internal.h header).
 *
 * GF(2^255-19) in radix 2^51.  Restates what curve25519-dalek 2.x `FieldElement51` [3P, not under
 * /root/reference] provides to the reference's call sites (e.g. every RistrettoPoint op in
 * src/nizk/presentation.rs:342-351).  Only canonical byte encodings are contractual
 * (SURVEY.md App. A.3), so the limb schedule is this file's own.
 */
#include "afx_oracle_internal.h"

#define M51 ((1ULL << 51) - 1)

void fe_0(fe* h) { memset(h, 0, sizeof *h); }
void fe_1(fe* h) { memset(h, 0, sizeof *h); h->v[0] = 1; }
void fe_copy(fe* h, const fe* f) { *h = *f; }

static void fe_weak_reduce(fe* h) {
  uint64_t c;
  c = h->v[0] >> 51; h->v[0] &= M51; h->v[1] += c;
  c = h->v[1] >> 51; h->v[1] &= M51; h->v[2] += c;
  c = h->v[2] >> 51; h->v[2] &= M51; h->v[3] += c;
  c = h->v[3] >> 51; h->v[3] &= M51; h->v[4] += c;
  c = h->v[4] >> 51; h->v[4] &= M51; h->v[0] += 19 * c;
}

void fe_add(fe* h, const fe* f, const fe* g) {
  for (int i = 0; i < 5; i++) h->v[i] = f->v[i] + g->v[i];
  fe_weak_reduce(h);
}

void fe_sub(fe* h, const fe* f, const fe* g) {
  /* add 16p so no limb underflows (g limbs < 2^52 after weak reduce) */
  h->v[0] = f->v[0] + 36028797018963664ULL - g->v[0];
  h->v[1] = f->v[1] + 36028797018963952ULL - g->v[1];
  h->v[2] = f->v[2] + 36028797018963952ULL - g->v[2];
  h->v[3] = f->v[3] + 36028797018963952ULL - g->v[3];
  h->v[4] = f->v[4] + 36028797018963952ULL - g->v[4];
  fe_weak_reduce(h);
}

void fe_neg(fe* h, const fe* f) {
  fe z; fe_0(&z);
  fe_sub(h, &z, f);
}

void fe_mul(fe* h, const fe* f, const fe* g) {
  const uint64_t f0 = f->v[0], f1 = f->v[1], f2 = f->v[2], f3 = f->v[3], f4 = f->v[4];
  const uint64_t g0 = g->v[0], g1 = g->v[1], g2 = g->v[2], g3 = g->v[3], g4 = g->v[4];
  const uint64_t g1_19 = 19 * g1, g2_19 = 19 * g2, g3_19 = 19 * g3, g4_19 = 19 * g4;
  u128 c0 = (u128)f0 * g0 + (u128)f1 * g4_19 + (u128)f2 * g3_19 + (u128)f3 * g2_19 + (u128)f4 * g1_19;
  u128 c1 = (u128)f0 * g1 + (u128)f1 * g0 + (u128)f2 * g4_19 + (u128)f3 * g3_19 + (u128)f4 * g2_19;
  u128 c2 = (u128)f0 * g2 + (u128)f1 * g1 + (u128)f2 * g0 + (u128)f3 * g4_19 + (u128)f4 * g3_19;
  u128 c3 = (u128)f0 * g3 + (u128)f1 * g2 + (u128)f2 * g1 + (u128)f3 * g0 + (u128)f4 * g4_19;
  u128 c4 = (u128)f0 * g4 + (u128)f1 * g3 + (u128)f2 * g2 + (u128)f3 * g1 + (u128)f4 * g0;
  c1 += (uint64_t)(c0 >> 51); uint64_t r0 = (uint64_t)c0 & M51;
  c2 += (uint64_t)(c1 >> 51); uint64_t r1 = (uint64_t)c1 & M51;
  c3 += (uint64_t)(c2 >> 51); uint64_t r2 = (uint64_t)c2 & M51;
  c4 += (uint64_t)(c3 >> 51); uint64_t r3 = (uint64_t)c3 & M51;
  uint64_t carry = (uint64_t)(c4 >> 51); uint64_t r4 = (uint64_t)c4 & M51;
  r0 += carry * 19;
  r1 += r0 >> 51; r0 &= M51;
  h->v[0] = r0; h->v[1] = r1; h->v[2] = r2; h->v[3] = r3; h->v[4] = r4;
}

void fe_sq(fe* h, const fe* f) { fe_mul(h, f, f); }

void fe_sqn(fe* h, const fe* f, int n) {
  fe_sq(h, f);
  for (int i = 1; i < n; i++) fe_sq(h, h);
}

void fe_frombytes(fe* h, const uint8_t s[32]) {
  uint64_t w[4];
  for (int i = 0; i < 4; i++) {
    w[i] = 0;
    for (int j = 0; j < 8; j++) w[i] |= (uint64_t)s[8 * i + j] << (8 * j);
  }
  h->v[0] = w[0] & M51;
  h->v[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
  h->v[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
  h->v[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
  h->v[4] = (w[3] >> 12) & M51; /* drops bit 255 */
}

void fe_tobytes(uint8_t s[32], const fe* f) {
  fe t = *f;
  fe_weak_reduce(&t);
  fe_weak_reduce(&t);
  /* now t < 2^255 + small; compute q = floor((t + 19) / 2^255) and subtract q*p */
  uint64_t q = (t.v[0] + 19) >> 51;
  q = (t.v[1] + q) >> 51;
  q = (t.v[2] + q) >> 51;
  q = (t.v[3] + q) >> 51;
  q = (t.v[4] + q) >> 51;
  t.v[0] += 19 * q;
  uint64_t c;
  c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
  c = t.v[1] >> 51; t.v[1] &= M51; t.v[2] += c;
  c = t.v[2] >> 51; t.v[2] &= M51; t.v[3] += c;
  c = t.v[3] >> 51; t.v[3] &= M51; t.v[4] += c;
  t.v[4] &= M51;
  uint64_t w[4];
  w[0] = t.v[0] | (t.v[1] << 51);
  w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
  w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
  w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) s[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}

/* z^(2^250-1) and z^11, shared by invert / pow22523 */
static void fe_pow22501(fe* t19, fe* t3, const fe* z) {
  fe t0, t1, t2, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15, t16, t17, t18;
  fe_sq(&t0, z);            /* 2 */
  fe_sqn(&t1, &t0, 2);      /* 8 */
  fe_mul(&t2, z, &t1);      /* 9 */
  fe_mul(t3, &t0, &t2);     /* 11 */
  fe_sq(&t4, t3);           /* 22 */
  fe_mul(&t5, &t2, &t4);    /* 31 = 2^5-1 */
  fe_sqn(&t6, &t5, 5);
  fe_mul(&t7, &t6, &t5);    /* 2^10-1 */
  fe_sqn(&t8, &t7, 10);
  fe_mul(&t9, &t8, &t7);    /* 2^20-1 */
  fe_sqn(&t10, &t9, 20);
  fe_mul(&t11, &t10, &t9);  /* 2^40-1 */
  fe_sqn(&t12, &t11, 10);
  fe_mul(&t13, &t12, &t7);  /* 2^50-1 */
  fe_sqn(&t14, &t13, 50);
  fe_mul(&t15, &t14, &t13); /* 2^100-1 */
  fe_sqn(&t16, &t15, 100);
  fe_mul(&t17, &t16, &t15); /* 2^200-1 */
  fe_sqn(&t18, &t17, 50);
  fe_mul(t19, &t18, &t13);  /* 2^250-1 */
}

void fe_invert(fe* out, const fe* z) {
  fe t19, t3, t20;
  fe_pow22501(&t19, &t3, z);
  fe_sqn(&t20, &t19, 5);    /* 2^255 - 32 */
  fe_mul(out, &t20, &t3);   /* 2^255 - 21 */
}

void fe_pow22523(fe* out, const fe* z) {
  fe t19, t3, t20;
  fe_pow22501(&t19, &t3, z);
  fe_sqn(&t20, &t19, 2);    /* 2^252 - 4 */
  fe_mul(out, z, &t20);     /* 2^252 - 3 */
}

int fe_is_negative(const fe* f) {
  uint8_t s[32];
  fe_tobytes(s, f);
  return s[0] & 1;
}

int fe_is_zero(const fe* f) {
  uint8_t s[32];
  fe_tobytes(s, f);
  uint8_t r = 0;
  for (int i = 0; i < 32; i++) r |= s[i];
  return r == 0;
}

int fe_eq(const fe* f, const fe* g) {
  uint8_t a[32], b[32];
  fe_tobytes(a, f);
  fe_tobytes(b, g);
  return memcmp(a, b, 32) == 0;
}

void fe_cmov(fe* f, const fe* g, int b) {
  uint64_t m = (uint64_t)0 - (uint64_t)(b & 1);
  for (int i = 0; i < 5; i++) f->v[i] ^= m & (f->v[i] ^ g->v[i]);
}

void fe_cneg(fe* f, int b) {
  fe n;
  fe_neg(&n, f);
  fe_cmov(f, &n, b);
}

void fe_abs(fe* h, const fe* f) {
  *h = *f;
  fe_cneg(h, fe_is_negative(f));
}

/* RFC 9496 §4.2 SQRT_RATIO_M1(u, v) == dalek FieldElement::sqrt_ratio_i */
int fe_sqrt_ratio_i(fe* r_out, const fe* u, const fe* v) {
  fe v3, v7, r, check, t, neg_u, neg_u_i, r_prime;
  fe_sq(&t, v);
  fe_mul(&v3, &t, v);           /* v^3 */
  fe_sq(&t, &v3);
  fe_mul(&v7, &t, v);           /* v^7 */
  fe_mul(&t, u, &v7);
  fe_pow22523(&t, &t);          /* (u v^7)^((p-5)/8) */
  fe_mul(&r, u, &v3);
  fe_mul(&r, &r, &t);           /* r = u v^3 (u v^7)^((p-5)/8) */
  fe_sq(&t, &r);
  fe_mul(&check, v, &t);        /* v r^2 */
  fe_neg(&neg_u, u);
  fe_mul(&neg_u_i, &neg_u, &FE_SQRT_M1);
  int correct_sign = fe_eq(&check, u);
  int flipped_sign = fe_eq(&check, &neg_u);
  int flipped_sign_i = fe_eq(&check, &neg_u_i);
  fe_mul(&r_prime, &r, &FE_SQRT_M1);
  fe_cmov(&r, &r_prime, flipped_sign | flipped_sign_i);
  fe_abs(&r, &r);
  *r_out = r;
  return correct_sign | flipped_sign;
}

fe FE_D, FE_D2, FE_SQRT_M1, FE_SQRT_AD_MINUS_ONE, FE_INVSQRT_A_MINUS_D, FE_ONE_MINUS_D_SQ, FE_D_MINUS_ONE_SQ;

#include "constants.inc"

void afxo_init_constants(void) {
  static int done = 0;
  if (done) return;
  fe_frombytes(&FE_D, K_D);
  fe_frombytes(&FE_D2, K_D2);
  fe_frombytes(&FE_SQRT_M1, K_SQRT_M1);
  fe_frombytes(&FE_SQRT_AD_MINUS_ONE, K_SQRT_AD_MINUS_ONE);
  fe_frombytes(&FE_INVSQRT_A_MINUS_D, K_INVSQRT_A_MINUS_D);
  fe_frombytes(&FE_ONE_MINUS_D_SQ, K_ONE_MINUS_D_SQ);
  fe_frombytes(&FE_D_MINUS_ONE_SQ, K_D_MINUS_ONE_SQ);
  done = 1;
}

void afxo_base_xy(fe* x, fe* y) {
  fe_frombytes(x, K_BASE_X);
  fe_frombytes(y, K_BASE_Y);
}
