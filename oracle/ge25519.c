/* ORACLE — test infrastructure only (see afx_oracle_internal.h header).
 *
 * edwards25519 in extended coordinates and the ristretto255 group (RFC 9496), restating what
 * curve25519-dalek 2.x `RistrettoPoint` / `CompressedRistretto` [3P] provide to the reference:
 * compress/decompress (src/parameters.rs:80, src/nizk/presentation.rs:373-412),
 * from_uniform_bytes / random (src/amacs.rs:290), point +,-,* (src/nizk/presentation.rs:342-351),
 * multiscalar_mul (src/amacs.rs:270) and vartime_multiscalar_mul (inside zkp verify_compact).
 * Only group elements / canonical encodings are contractual, not the addition schedule.
 */
#include "afx_oracle_internal.h"

void afxo_base_xy(fe* x, fe* y);

void ge_identity(ge* h) {
  fe_0(&h->X); fe_1(&h->Y); fe_1(&h->Z); fe_0(&h->T);
}

/* add-2008-hwcd-3 (a = -1), unified */
void ge_add(ge* r, const ge* p, const ge* q) {
  fe a, b, c, d, e, f, g, h, t0, t1;
  fe_sub(&t0, &p->Y, &p->X);
  fe_sub(&t1, &q->Y, &q->X);
  fe_mul(&a, &t0, &t1);
  fe_add(&t0, &p->Y, &p->X);
  fe_add(&t1, &q->Y, &q->X);
  fe_mul(&b, &t0, &t1);
  fe_mul(&c, &p->T, &q->T);
  fe_mul(&c, &c, &FE_D2);
  fe_mul(&d, &p->Z, &q->Z);
  fe_add(&d, &d, &d);
  fe_sub(&e, &b, &a);
  fe_sub(&f, &d, &c);
  fe_add(&g, &d, &c);
  fe_add(&h, &b, &a);
  fe_mul(&r->X, &e, &f);
  fe_mul(&r->Y, &g, &h);
  fe_mul(&r->T, &e, &h);
  fe_mul(&r->Z, &f, &g);
}

void ge_neg(ge* r, const ge* p) {
  fe_neg(&r->X, &p->X);
  r->Y = p->Y;
  r->Z = p->Z;
  fe_neg(&r->T, &p->T);
}

void ge_sub(ge* r, const ge* p, const ge* q) {
  ge nq;
  ge_neg(&nq, q);
  ge_add(r, p, &nq);
}

/* dbl-2008-hwcd (a = -1) */
void ge_double(ge* r, const ge* p) {
  fe a, b, c, d, e, f, g, h, t0;
  fe_sq(&a, &p->X);
  fe_sq(&b, &p->Y);
  fe_sq(&c, &p->Z);
  fe_add(&c, &c, &c);
  fe_neg(&d, &a);
  fe_add(&t0, &p->X, &p->Y);
  fe_sq(&e, &t0);
  fe_sub(&e, &e, &a);
  fe_sub(&e, &e, &b);
  fe_add(&g, &d, &b);
  fe_sub(&f, &g, &c);
  fe_sub(&h, &d, &b);
  fe_mul(&r->X, &e, &f);
  fe_mul(&r->Y, &g, &h);
  fe_mul(&r->T, &e, &h);
  fe_mul(&r->Z, &f, &g);
}

/* signed radix-16 digits, 64 of them, each in [-8, 8] */
static void sc_radix16(int8_t e[64], const sc* s) {
  for (int i = 0; i < 32; i++) {
    e[2 * i] = s->b[i] & 15;
    e[2 * i + 1] = (s->b[i] >> 4) & 15;
  }
  int8_t carry = 0;
  for (int i = 0; i < 63; i++) {
    e[i] += carry;
    carry = (int8_t)((e[i] + 8) >> 4);
    e[i] -= (int8_t)(carry << 4);
  }
  e[63] += carry;
}

static void table8(ge t[8], const ge* p) {
  t[0] = *p;
  for (int i = 1; i < 8; i++) ge_add(&t[i], &t[i - 1], p);
}

static void add_digit(ge* r, const ge t[8], int d) {
  if (d > 0) ge_add(r, r, &t[d - 1]);
  else if (d < 0) ge_sub(r, r, &t[-d - 1]);
}

void ge_multiscalar(ge* r, const sc* s, const ge* p, int n) {
  ge tab[ZKP_MAX_TERMS][8];
  int8_t e[ZKP_MAX_TERMS][64];
  for (int k = 0; k < n; k++) { table8(tab[k], &p[k]); sc_radix16(e[k], &s[k]); }
  ge acc;
  ge_identity(&acc);
  for (int i = 63; i >= 0; i--) {
    if (i != 63) { ge_double(&acc, &acc); ge_double(&acc, &acc); ge_double(&acc, &acc); ge_double(&acc, &acc); }
    for (int k = 0; k < n; k++) add_digit(&acc, tab[k], e[k][i]);
  }
  *r = acc;
}

void ge_scalarmult(ge* r, const sc* s, const ge* p) { ge_multiscalar(r, s, p, 1); }

/* width-5 non-adjacent form, digits odd in [-15, 15] */
static void sc_naf5(int8_t naf[256], const sc* s) {
  uint64_t x[5] = {0};
  for (int i = 0; i < 32; i++) x[i / 8] |= (uint64_t)s->b[i] << (8 * (i % 8));
  memset(naf, 0, 256);
  int pos = 0, carry = 0;
  while (pos < 256) {
    int idx = pos / 64, bit = pos % 64;
    uint64_t buf = (bit < 59) ? (x[idx] >> bit) : ((x[idx] >> bit) | (x[idx + 1] << (64 - bit)));
    int window = carry + (int)(buf & 31);
    if ((window & 1) == 0) { pos += 1; continue; }
    if (window < 16) { carry = 0; naf[pos] = (int8_t)window; }
    else { carry = 1; naf[pos] = (int8_t)(window - 32); }
    pos += 5;
  }
}

void ge_multiscalar_vartime(ge* r, const sc* s, const ge* p, int n) {
  ge tab[ZKP_MAX_TERMS][8]; /* odd multiples 1,3,...,15 */
  int8_t naf[ZKP_MAX_TERMS][256];
  for (int k = 0; k < n; k++) {
    ge p2;
    ge_double(&p2, &p[k]);
    tab[k][0] = p[k];
    for (int i = 1; i < 8; i++) ge_add(&tab[k][i], &tab[k][i - 1], &p2);
    sc_naf5(naf[k], &s[k]);
  }
  ge acc;
  ge_identity(&acc);
  int started = 0;
  for (int i = 255; i >= 0; i--) {
    if (started) ge_double(&acc, &acc);
    for (int k = 0; k < n; k++) {
      int d = naf[k][i];
      if (d > 0) { ge_add(&acc, &acc, &tab[k][d >> 1]); started = 1; }
      else if (d < 0) { ge_sub(&acc, &acc, &tab[k][(-d) >> 1]); started = 1; }
    }
  }
  *r = acc;
}

/* RFC 9496 §4.3.1 Decode */
int ristretto_decode(ge* r, const uint8_t in[32]) {
  fe s, ss, u1, u2, u2s, v, t, I, Dx, Dy, x, y, one;
  uint8_t chk[32];
  fe_frombytes(&s, in);
  fe_tobytes(chk, &s);
  if (memcmp(chk, in, 32) != 0) return 0;   /* non-canonical */
  if (fe_is_negative(&s)) return 0;
  fe_1(&one);
  fe_sq(&ss, &s);
  fe_sub(&u1, &one, &ss);
  fe_add(&u2, &one, &ss);
  fe_sq(&u2s, &u2);
  fe_mul(&v, &FE_D, &u1);
  fe_mul(&v, &v, &u1);
  fe_neg(&v, &v);
  fe_sub(&v, &v, &u2s);                     /* v = -(D u1^2) - u2^2 */
  fe_mul(&t, &v, &u2s);
  int was_square = fe_sqrt_ratio_i(&I, &one, &t);
  fe_mul(&Dx, &I, &u2);
  fe_mul(&Dy, &I, &Dx);
  fe_mul(&Dy, &Dy, &v);
  fe_add(&x, &s, &s);
  fe_mul(&x, &x, &Dx);
  fe_abs(&x, &x);
  fe_mul(&y, &u1, &Dy);
  fe_mul(&t, &x, &y);
  if (!was_square || fe_is_negative(&t) || fe_is_zero(&y)) return 0;
  r->X = x; r->Y = y; fe_1(&r->Z); r->T = t;
  return 1;
}

/* RFC 9496 §4.3.2 Encode */
void ristretto_encode(uint8_t out[32], const ge* p) {
  fe u1, u2, t, I, D1, D2, Zinv, ix, iy, ead, x, y, z, Dinv, s;
  fe_add(&u1, &p->Z, &p->Y);
  fe_sub(&t, &p->Z, &p->Y);
  fe_mul(&u1, &u1, &t);
  fe_mul(&u2, &p->X, &p->Y);
  fe_sq(&t, &u2);
  fe_mul(&t, &t, &u1);
  fe one; fe_1(&one);
  fe_sqrt_ratio_i(&I, &one, &t);
  fe_mul(&D1, &u1, &I);
  fe_mul(&D2, &u2, &I);
  fe_mul(&Zinv, &D1, &D2);
  fe_mul(&Zinv, &Zinv, &p->T);
  fe_mul(&ix, &p->X, &FE_SQRT_M1);
  fe_mul(&iy, &p->Y, &FE_SQRT_M1);
  fe_mul(&ead, &D1, &FE_INVSQRT_A_MINUS_D);
  fe_mul(&t, &p->T, &Zinv);
  int rotate = fe_is_negative(&t);
  x = p->X; y = p->Y; Dinv = D2;
  fe_cmov(&x, &iy, rotate);
  fe_cmov(&y, &ix, rotate);
  z = p->Z;
  fe_cmov(&Dinv, &ead, rotate);
  fe_mul(&t, &x, &Zinv);
  fe_cneg(&y, fe_is_negative(&t));
  fe_sub(&s, &z, &y);
  fe_mul(&s, &s, &Dinv);
  fe_abs(&s, &s);
  fe_tobytes(out, &s);
}

/* RFC 9496 §4.3.4 MAP (dalek RistrettoPoint::elligator_ristretto_flavor) */
static void elligator(ge* out, const fe* r0) {
  fe r, u, v, s, s_prime, c, N, w0, w1, w2, w3, one, t, rp1;
  fe_1(&one);
  fe_sq(&r, r0);
  fe_mul(&r, &r, &FE_SQRT_M1);
  fe_add(&rp1, &r, &one);
  fe_mul(&u, &rp1, &FE_ONE_MINUS_D_SQ);
  fe_neg(&c, &one);
  fe_mul(&t, &r, &FE_D);
  fe_sub(&v, &c, &t);
  fe_add(&t, &r, &FE_D);
  fe_mul(&v, &v, &t);
  int was_square = fe_sqrt_ratio_i(&s, &u, &v);
  fe_mul(&s_prime, &s, r0);
  fe_abs(&s_prime, &s_prime);
  fe_neg(&s_prime, &s_prime);
  fe_cmov(&s, &s_prime, !was_square);
  fe_cmov(&c, &r, !was_square);
  fe_sub(&t, &r, &one);
  fe_mul(&N, &c, &t);
  fe_mul(&N, &N, &FE_D_MINUS_ONE_SQ);
  fe_sub(&N, &N, &v);
  fe_add(&w0, &s, &s);
  fe_mul(&w0, &w0, &v);
  fe_mul(&w1, &N, &FE_SQRT_AD_MINUS_ONE);
  fe_sq(&t, &s);
  fe_sub(&w2, &one, &t);
  fe_add(&w3, &one, &t);
  fe_mul(&out->X, &w0, &w3);
  fe_mul(&out->Y, &w2, &w1);
  fe_mul(&out->Z, &w1, &w3);
  fe_mul(&out->T, &w0, &w2);
}

void ristretto_from_uniform_bytes(ge* r, const uint8_t b[64]) {
  fe r1, r2;
  ge p1, p2;
  fe_frombytes(&r1, b);
  fe_frombytes(&r2, b + 32);
  elligator(&p1, &r1);
  elligator(&p2, &r2);
  ge_add(r, &p1, &p2);
}

int ristretto_eq(const ge* p, const ge* q) {
  fe a, b;
  fe_mul(&a, &p->X, &q->Y);
  fe_mul(&b, &p->Y, &q->X);
  int e1 = fe_eq(&a, &b);
  fe_mul(&a, &p->X, &q->X);
  fe_mul(&b, &p->Y, &q->Y);
  int e2 = fe_eq(&a, &b);
  return e1 | e2;
}

void ristretto_basepoint(ge* r) {
  afxo_base_xy(&r->X, &r->Y);
  fe_1(&r->Z);
  fe_mul(&r->T, &r->X, &r->Y);
}
