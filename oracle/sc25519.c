/* ORACLE — test infrastructure only (see afx_oracle_internal.h header).
 *
 * Scalars mod l = 2^252 + 27742317777372353535851937790883648493.  Restates curve25519-dalek 2.x
 * `Scalar` [3P] as used by the reference: Scalar::random / from_bytes_mod_order_wide
 * (src/amacs.rs:93-102,289), negation and products (src/nizk/presentation.rs:163,
 * src/nizk/encryption.rs:78), from_canonical_bytes (src/amacs.rs:141-149).
 * Reduction uses 2^252 = -delta (mod l) folding; clarity over speed.
 */
#include "afx_oracle_internal.h"
#include "constants.inc"

/* little-endian multi-precision helpers on 64-bit limbs */
static void load_le(uint64_t* w, const uint8_t* b, int nbytes, int nlimbs) {
  for (int i = 0; i < nlimbs; i++) w[i] = 0;
  for (int i = 0; i < nbytes; i++) w[i / 8] |= (uint64_t)b[i] << (8 * (i % 8));
}
static void store_le32(uint8_t* b, const uint64_t* w) {
  for (int i = 0; i < 32; i++) b[i] = (uint8_t)(w[i / 8] >> (8 * (i % 8)));
}
/* r[0..n) = a[0..n) + b[0..n), returns carry */
static uint64_t mp_add(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) {
  u128 c = 0;
  for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
/* r = a - b, returns borrow (1 if a < b) */
static uint64_t mp_sub(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) {
  uint64_t borrow = 0;
  for (int i = 0; i < n; i++) {
    u128 t = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)t;
    borrow = (uint64_t)(t >> 64) & 1;
  }
  return borrow;
}
/* r[0..na+nb) = a * b */
static void mp_mul(uint64_t* r, const uint64_t* a, int na, const uint64_t* b, int nb) {
  for (int i = 0; i < na + nb; i++) r[i] = 0;
  for (int i = 0; i < na; i++) {
    u128 c = 0;
    for (int j = 0; j < nb; j++) {
      c += (u128)a[i] * b[j] + r[i + j];
      r[i + j] = (uint64_t)c;
      c >>= 64;
    }
    r[i + nb] = (uint64_t)c;
  }
}
/* split x (n limbs) at bit 252: lo = x mod 2^252 (4 limbs), hi = x >> 252 (n-3 limbs, zero padded) */
static void split252(uint64_t lo[4], uint64_t* hi, int nhi, const uint64_t* x, int n) {
  for (int i = 0; i < 4; i++) lo[i] = i < n ? x[i] : 0;
  lo[3] &= (1ULL << 60) - 1;
  for (int i = 0; i < nhi; i++) {
    uint64_t a = (i + 3 < n) ? x[i + 3] : 0, b = (i + 4 < n) ? x[i + 4] : 0;
    hi[i] = (a >> 60) | (b << 4);
  }
}

static const uint64_t DELTA[2] = { 0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL };
static const uint64_t LIMB_L[4] = { 0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL };

/* reduce an 8-limb (512-bit) value mod l */
static void reduce512(uint64_t r[4], const uint64_t x[8]) {
  uint64_t lo1[4], hi1[5], t1[7], lo2[4], hi2[4], t2[6], lo3[4], hi3[3], t3[5];
  split252(lo1, hi1, 5, x, 8);              /* hi1 < 2^260 */
  mp_mul(t1, hi1, 5, DELTA, 2);             /* < 2^385 */
  split252(lo2, hi2, 4, t1, 7);             /* hi2 < 2^133 */
  mp_mul(t2, hi2, 4, DELTA, 2);             /* < 2^258 (6 limbs) */
  split252(lo3, hi3, 3, t2, 6);             /* hi3 < 2^6 */
  mp_mul(t3, hi3, 3, DELTA, 2);             /* < 2^131 */
  /* x = lo1 - lo2 + lo3 - t3  (mod l); every term < 2^252, so the sum lies in (-2^253, 2^253) */
  uint64_t acc[5], tmp[5], l2[5];
  for (int i = 0; i < 4; i++) { acc[i] = lo1[i]; tmp[i] = lo3[i]; }
  acc[4] = tmp[4] = 0;
  mp_add(acc, acc, tmp, 5);
  /* + 2l to stay non-negative */
  for (int i = 0; i < 4; i++) l2[i] = LIMB_L[i];
  l2[4] = 0;
  mp_add(acc, acc, l2, 5);
  mp_add(acc, acc, l2, 5);
  for (int i = 0; i < 4; i++) tmp[i] = lo2[i];
  tmp[4] = 0;
  mp_sub(acc, acc, tmp, 5);
  for (int i = 0; i < 5; i++) tmp[i] = t3[i];
  tmp[4] = 0; /* t3 < 2^131 fits 3 limbs */
  mp_sub(acc, acc, tmp, 5);
  /* acc in [0, 2^254 + 2l): subtract l while >= l (at most 5 times) */
  for (int k = 0; k < 6; k++) {
    uint64_t s[5];
    uint64_t borrow = mp_sub(s, acc, l2, 5);
    if (!borrow) memcpy(acc, s, sizeof s);
  }
  for (int i = 0; i < 4; i++) r[i] = acc[i];
}

void sc_reduce_wide(sc* r, const uint8_t in[64]) {
  uint64_t x[8], o[4];
  load_le(x, in, 64, 8);
  reduce512(o, x);
  store_le32(r->b, o);
}

void sc_from_bytes_mod_order(sc* r, const uint8_t in[32]) {
  uint8_t w[64];
  memcpy(w, in, 32);
  memset(w + 32, 0, 32);
  sc_reduce_wide(r, w);
}

int sc_is_canonical(const uint8_t in[32]) {
  /* in < l ? compare big-endian-wise from the top byte */
  for (int i = 31; i >= 0; i--) {
    if (in[i] < K_L[i]) return 1;
    if (in[i] > K_L[i]) return 0;
  }
  return 0;
}

void sc_zero(sc* r) { memset(r->b, 0, 32); }
void sc_one(sc* r) { memset(r->b, 0, 32); r->b[0] = 1; }
int sc_eq(const sc* a, const sc* b) { return memcmp(a->b, b->b, 32) == 0; }

void sc_mul(sc* r, const sc* a, const sc* b) {
  uint64_t x[4], y[4], p[8], o[4];
  load_le(x, a->b, 32, 4);
  load_le(y, b->b, 32, 4);
  mp_mul(p, x, 4, y, 4);
  reduce512(o, p);
  store_le32(r->b, o);
}

void sc_add(sc* r, const sc* a, const sc* b) {
  uint64_t x[8] = {0}, y[8] = {0}, o[4];
  load_le(x, a->b, 32, 4);
  load_le(y, b->b, 32, 4);
  x[4] = mp_add(x, x, y, 4);
  reduce512(o, x);
  store_le32(r->b, o);
}

void sc_neg(sc* r, const sc* a) {
  uint64_t x[8] = {0}, o[4], t[4];
  sc ar;
  sc_from_bytes_mod_order(&ar, a->b);
  load_le(t, ar.b, 32, 4);
  mp_sub(x, LIMB_L, t, 4); /* l - a in (0, l] */
  reduce512(o, x);
  store_le32(r->b, o);
}

void sc_sub(sc* r, const sc* a, const sc* b) {
  sc nb;
  sc_neg(&nb, b);
  sc_add(r, a, &nb);
}

void sc_muladd(sc* r, const sc* a, const sc* b, const sc* c) {
  sc t;
  sc_mul(&t, a, b);
  sc_add(r, &t, c);
}
