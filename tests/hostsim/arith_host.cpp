// The engine's DEVICE arithmetic headers (fe.cuh, sc.cuh, ge.cuh), compiled for the host so that the CPU test-suite
// can check them against the oracle without a GPU, and so that the field-operation counts DESIGN.md publishes are
// measured rather than estimated.  Test infrastructure only: nothing in the product links this file.
#include <stdint.h>
#include <string.h>
thread_local uint64_t afx_n_mul = 0, afx_n_sq = 0, afx_n_chain_mul = 0, afx_n_chain_sq = 0;
#define AFX_COUNT_OPS 1
// every fe_mul / fe_sq of this build checks its operand bounds on the actual values (fe.cuh, AFX_CHECK_BOUNDS)
#define AFX_CHECK_BOUNDS 1
static uint64_t n_violations = 0;
static const char* last_violation = "";
extern "C" void afx_bounds_violation(const char* what) { n_violations++; last_violation = what; }
#include "../../aeonflux_amd/csrc/ge.cuh"
#include "../../aeonflux_amd/csrc/sc.cuh"

static void load8(uint32_t w[8], const uint8_t* p) { memcpy(w, p, 32); }
static sc sc_from(const uint8_t* p) { sc s; memcpy(s.v, p, 32); return s; }

extern "C" {

uint64_t arith_bounds_violations(void) { return n_violations; }
void arith_reset_violations(uint64_t to) { n_violations = to; }   // for tests that feed out-of-bounds operands on purpose
// self-test of the checker: operands of three units' magnitude (column sums beyond 2^63; a doubled limb beyond int32) must
// trip it; returns the number of violations it caused and leaves the global count as it was
uint64_t arith_bounds_checker_selftest(void) {
  const uint64_t before = n_violations;
  fe big;
  for (int i = 0; i < AFX_FE_LIMBS; i++) big.v[i] = 3 << (i < 8 ? 29 : 23);
  (void)fe_mul(big, big);          // 8 * 9 * 2^58 > 2^63
  (void)fe_sq(big);                // 2 * 3 * 2^29 does not fit int32
  const uint64_t caused = n_violations - before;
  n_violations = before;
  return caused;
}
const char* arith_last_violation(void) { return last_violation; }

// The doubling/addition chain of k_msm, step for step, with the same consumer-aware conversions (ge.cuh GE_FOR_*):
// out = sum_t s[t] * P[t] (nv variable bases, signed 4-bit windows, per-"lane" tables as the kernel builds them)
//       + sum_u f[u] * Q[u] (nf points added as affine niels entries after the chain, bit by bit, standing in for the
//       positional tables).  Returns 0 if a point does not decode.
int arith_msm_chain(uint8_t out[32], uint32_t nv, const uint8_t* s, const uint8_t* p, uint32_t nf, const uint8_t* f, const uint8_t* q) {
  if (nv == 0 || nv > 8 || nf > 4) return 0;
  ge_cached tab[8][9];
  uint32_t digits[8][8];
  for (uint32_t t = 0; t < nv; t++) {
    uint32_t w[8];
    load8(w, p + 32 * t);
    ge_p3 P;
    if (!ristretto_decode(P, w)) return 0;
    const ge_cached cP = ge_p3_to_cached(P);   // as msm_build_table: the packing below is what reduces the entries
    tab[t][0] = ge_cached_identity();
    tab[t][1] = cP;
    ge_p3 Q = P;
    for (int k = 2; k < 9; k++) { Q = ge_p1p1_to_p3(ge_add_cached(Q, cP, false)); tab[t][k] = ge_p3_to_cached(Q); }
    // the kernel stores entries in canonical 32-byte form and reloads them: do the same round trip
    for (int k = 0; k < 9; k++) {
      uint32_t b[8];
      fe_tobytes(b, tab[t][k].YpX); tab[t][k].YpX = fe_frombytes(b);
      fe_tobytes(b, tab[t][k].YmX); tab[t][k].YmX = fe_frombytes(b);
      fe_tobytes(b, tab[t][k].Z2); tab[t][k].Z2 = fe_frombytes(b);
      fe_tobytes(b, tab[t][k].T2d); tab[t][k].T2d = fe_frombytes(b);
    }
    sc_bias(digits[t], sc_from(s + 32 * t), 0x88888888u);
  }
  ge_p3 acc = ge_identity();
  for (int w = 63; w >= 0; w--) {
    if (w != 63) {
      ge_p2 a2 = ge_p3_to_p2(acc);
      for (int k = 0; k < 3; k++) a2 = ge_p1p1_to_p2_before_dbl(ge_p2_dbl(a2));
      acc = ge_p1p1_to_p3_for<GE_FOR_ADD>(ge_p2_dbl(a2));
    }
    for (uint32_t t = 0; t < nv; t++) {
      const int d = (int)((digits[t][w >> 3] >> ((w & 7) * 4)) & 15u) - 8;
      const int idx = d < 0 ? -d : d;
      const int next = t + 1 < nv ? GE_FOR_ADD : (w == 0 ? GE_FOR_ANY : GE_FOR_DBL);
      acc = ge_p1p1_to_p3_next(ge_add_cached(acc, tab[t][idx], d < 0), next);
    }
  }
  // niels additions after the chain: f[u] * Q[u] bit by bit with 2^bit * Q[u] made affine on the fly
  for (uint32_t u = 0; u < nf; u++) {
    uint32_t w[8];
    load8(w, q + 32 * u);
    ge_p3 B;
    if (!ristretto_decode(B, w)) return 0;
    for (int bit = 0; bit < 253; bit++) {
      if ((f[32 * u + (bit >> 3)] >> (bit & 7)) & 1) {
        const fe zinv = fe_invert(B.Z);
        const ge_niels nq = ge_niels_from_affine(fe_mul(B.X, zinv), fe_mul(B.Y, zinv));
        const bool last = u + 1 == nf && bit == 252;
        acc = ge_p1p1_to_p3_next(ge_madd(acc, nq, false), last ? GE_FOR_ANY : GE_FOR_ADD);
      }
      B = ge_double(B);
    }
  }
  acc = ge_carry(acc);   // whatever the last consumer annotation was, the encoder takes any reduced point
  uint32_t o[8];
  ristretto_encode(o, acc);
  memcpy(out, o, 32);
  return 1;
}

void arith_counters(uint64_t out[2], int reset) {
  out[0] = afx_n_mul; out[1] = afx_n_sq;
  if (reset) afx_n_mul = afx_n_sq = 0;
}
// the share of the above executed inside the inversion / square-root chains (fe10.cuh: the 10-limb form); never reset by arith_counters
void arith_chain_counters(uint64_t out[2], int reset) {
  out[0] = afx_n_chain_mul; out[1] = afx_n_chain_sq;
  if (reset) afx_n_chain_mul = afx_n_chain_sq = 0;
}
// chain operations of one decode, one encode, one plain inversion (plan.h AFX_CHAIN_*)
void arith_chain_op_counts(uint64_t out[3][2]) {
  uint8_t enc[32] = { 0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                      0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76 };   // basepoint
  uint32_t w[8], o[8];
  load8(w, enc);
  ge_p3 P;
  uint64_t c[2];
  arith_chain_counters(c, 1);
  ristretto_decode(P, w); arith_chain_counters(out[0], 1);
  ristretto_encode(o, P); arith_chain_counters(out[1], 1);
  const fe inv = fe_invert(P.Z); arith_chain_counters(out[2], 1);
  (void)inv;
}

// out = canonical encoding of a*b, a^2, 1/a, a^((p-5)/8) over GF(2^255-19); inputs 32-byte little endian (bit 255 ignored)
void arith_fe(uint8_t out[4][32], const uint8_t a[32], const uint8_t b[32]) {
  uint32_t wa[8], wb[8], w[8];
  load8(wa, a); load8(wb, b);
  const fe x = fe_frombytes(wa), y = fe_frombytes(wb);
  fe_tobytes(w, fe_mul(x, y)); memcpy(out[0], w, 32);
  fe_tobytes(w, fe_sq(x)); memcpy(out[1], w, 32);
  fe_tobytes(w, fe_invert(x)); memcpy(out[2], w, 32);
  fe_tobytes(w, fe_pow22523(x)); memcpy(out[3], w, 32);
}

// lazily-added operands at the documented bounds: (a0 + a1 + a2 + a3) * (b0 - b1), and (a0 - a1)^2
void arith_fe_lazy(uint8_t out[2][32], const uint8_t a[4][32], const uint8_t b[2][32]) {
  fe x[4], y[2];
  uint32_t w[8];
  for (int i = 0; i < 4; i++) { load8(w, a[i]); x[i] = fe_carry(fe_frombytes(w)); }
  for (int i = 0; i < 2; i++) { load8(w, b[i]); y[i] = fe_carry(fe_frombytes(w)); }
  const fe s = fe_add(fe_add(x[0], x[1]), fe_add(x[2], x[3]));   // four reduced terms: fe_mul's first operand
  fe_tobytes(w, fe_mul(s, fe_sub(y[0], y[1]))); memcpy(out[0], w, 32);
  fe_tobytes(w, fe_sq(fe_sub(x[0], x[1]))); memcpy(out[1], w, 32);
}

// operands given limb by limb (any int32 values the caller likes): canonical bytes of f*g (raw and centred flavours), f^2
// (both), and of f itself; every product goes through the bounds checker
// which: 1 = the products, 2 = the squarings, 4 = f itself
void arith_fe_limbs(uint8_t out[5][32], const int32_t f[AFX_FE_LIMBS], const int32_t g[AFX_FE_LIMBS], int which) {
  fe x, y;
  for (int i = 0; i < AFX_FE_LIMBS; i++) { x.v[i] = f[i]; y.v[i] = g[i]; }
  uint32_t w[8];
  if (which & 1) {
    fe_tobytes(w, fe_mul(x, y)); memcpy(out[0], w, 32);
    fe_tobytes(w, fe_mul_raw(x, y)); memcpy(out[1], w, 32);
  }
  if (which & 2) {
    fe_tobytes(w, fe_sq(x)); memcpy(out[2], w, 32);
    fe_tobytes(w, fe_sq_raw(x)); memcpy(out[3], w, 32);
  }
  if (which & 4) { fe_tobytes(w, x); memcpy(out[4], w, 32); }
}
// limbs of a raw / centred product, for range checks
void arith_fe_mul_limbs(int32_t raw[AFX_FE_LIMBS], int32_t centred[AFX_FE_LIMBS], const int32_t f[AFX_FE_LIMBS], const int32_t g[AFX_FE_LIMBS]) {
  fe x, y;
  for (int i = 0; i < AFX_FE_LIMBS; i++) { x.v[i] = f[i]; y.v[i] = g[i]; }
  const fe r = fe_mul_raw(x, y), c = fe_mul(x, y);
  for (int i = 0; i < AFX_FE_LIMBS; i++) { raw[i] = r.v[i]; centred[i] = c.v[i]; }
}

int arith_decode_encode(uint8_t out[32], const uint8_t in[32]) {
  uint32_t w[8], o[8];
  load8(w, in);
  ge_p3 P;
  const bool ok = ristretto_decode(P, w);
  if (!ok) return 0;
  ristretto_encode(o, P);
  memcpy(out, o, 32);
  return 1;
}

// out = s*P by plain double-and-add over the engine's point formulas; add_out = P + Q, sub_out = P - Q, dbl_out = 2P
int arith_point_ops(uint8_t mul_out[32], uint8_t add_out[32], uint8_t sub_out[32], uint8_t dbl_out[32], const uint8_t s[32],
                    const uint8_t p[32], const uint8_t q[32]) {
  uint32_t w[8], o[8];
  ge_p3 P, Q;
  load8(w, p); if (!ristretto_decode(P, w)) return 0;
  load8(w, q); if (!ristretto_decode(Q, w)) return 0;
  ge_p3 acc = ge_identity();
  const ge_cached cP = ge_p3_to_cached(P);
  for (int bit = 255; bit >= 0; bit--) {
    acc = ge_double(acc);
    if ((s[bit >> 3] >> (bit & 7)) & 1) acc = ge_p1p1_to_p3(ge_add_cached(acc, cP, false));
  }
  ristretto_encode(o, acc); memcpy(mul_out, o, 32);
  ristretto_encode(o, ge_add(P, Q)); memcpy(add_out, o, 32);
  ristretto_encode(o, ge_sub(P, Q)); memcpy(sub_out, o, 32);
  ristretto_encode(o, ge_double(P)); memcpy(dbl_out, o, 32);
  return 1;
}

void arith_from_uniform(uint8_t out[32], const uint8_t wide[64]) {
  uint32_t w[16], o[8];
  memcpy(w, wide, 64);
  ristretto_encode(o, ristretto_from_uniform(w));
  memcpy(out, o, 32);
}

// scalars: reduce a 64-byte value; a*b + c; -a; canonicity of a
void arith_sc(uint8_t red[32], uint8_t muladd[32], uint8_t neg[32], int* canonical, const uint8_t wide[64], const uint8_t a[32],
              const uint8_t b[32], const uint8_t c[32]) {
  uint32_t x[16];
  memcpy(x, wide, 64);
  sc r = sc_reduce512(x); memcpy(red, r.v, 32);
  r = sc_muladd(sc_from(a), sc_from(b), sc_from(c)); memcpy(muladd, r.v, 32);
  r = sc_neg(sc_from(a)); memcpy(neg, r.v, 32);
  *canonical = sc_is_canonical(sc_from(a)) ? 1 : 0;
}

// s/2 and 2s mod l (k_msm's recoding of halved jobs and of terms on a half base)
void arith_sc_half_dbl(uint8_t half[32], uint8_t dbl[32], const uint8_t a[32]) {
  sc r = sc_half(sc_from(a)); memcpy(half, r.v, 32);
  r = sc_dbl(sc_from(a)); memcpy(dbl, r.v, 32);
}

// signed-digit recoding used by k_msm: digits of s + bias; returns sum(d_i * radix^i) check value via the caller
void arith_sc_bias(uint32_t out[8], const uint8_t s[32], uint32_t bias) { sc_bias(out, sc_from(s), bias); }

// field-operation counts (mul, sq) of the building blocks of k_msm / k_decode, measured on this build
void arith_op_counts(uint64_t out[8][2]) {
  uint8_t enc[32] = { 0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                      0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76 };   // basepoint
  uint32_t w[8], o[8];
  load8(w, enc);
  ge_p3 P;
  uint64_t c[2];
  arith_counters(c, 1);
  ristretto_decode(P, w); arith_counters(out[0], 1);                                     // decode
  ristretto_encode(o, P); arith_counters(out[1], 1);                                     // encode
  ge_p2 p2 = ge_p1p1_to_p2(ge_p2_dbl(ge_p3_to_p2(P))); arith_counters(out[2], 1);        // doubling -> p2
  ge_p3 p3 = ge_p1p1_to_p3(ge_p2_dbl(p2)); arith_counters(out[3], 1);                    // doubling -> p3
  const ge_cached cP = ge_p3_to_cached_reduced(P); arith_counters(out[4], 1);            // p3 -> table entry
  p3 = ge_p1p1_to_p3(ge_add_cached(p3, cP, false)); arith_counters(out[5], 1);           // + variable-base entry -> p3
  ge_niels nq = ge_niels_identity();
  p3 = ge_p1p1_to_p3(ge_madd(p3, nq, false)); arith_counters(out[6], 1);                 // + generator entry -> p3
  p2 = ge_p1p1_to_p2(ge_add_cached(p3, cP, false)); arith_counters(out[7], 1);           // + variable-base entry -> p2
  (void)p2;
}

// double-and-compress (ge.cuh c2x_*; the engine's k_compress2x): encodings of 2*P_j for n points given by their encodings, with
// ONE inversion, every multiplication bound-checked.  Each P_j goes through a doubling and an addition first, so that the
// state is fed the kind of limbs an MSM accumulator holds, not freshly decoded ones.  Returns 0 if a point does not decode.
int arith_double_and_compress(uint8_t* out /* [n][32] */, uint32_t n, const uint8_t* pts /* [n][32] */, const uint8_t* shift /* 32 B point added to each */) {
  if (n > 64) return 0;
  c2x_state st[64];
  fe pre[64], prod = fe_one();
  uint32_t w[8];
  load8(w, shift);
  ge_p3 S;
  if (!ristretto_decode(S, w)) return 0;
  for (uint32_t j = 0; j < n; j++) {
    load8(w, pts + 32 * j);
    ge_p3 P;
    if (!ristretto_decode(P, w)) return 0;
    // (P + S) - S = P through the engine's own addition: limbs as an accumulator leaves them (ge_p1p1_to_p3: centred)
    P = ge_sub(ge_add(P, S), S);
    st[j] = c2x_from(P);
    pre[j] = prod;
    prod = fe_mul(prod, st[j].efgh);
  }
  fe inv = fe_invert(prod);
  for (uint32_t jj = n; jj > 0; jj--) {
    const uint32_t j = jj - 1;
    const fe inv_j = fe_mul(inv, pre[j]);
    inv = fe_mul(inv, st[j].efgh);
    c2x_finish(w, st[j], inv_j);
    memcpy(out + 32 * j, w, 32);
  }
  return 1;
}

// encodings of the NEGATIONS of n decoded points with one inversion (ge.cuh negenc_*; the engine's k_negenc), every multiplication
// bound-checked.  A point with a zero factor (the identity) is left out of the product and encodes to zeros.  Returns 0 if a
// point does not decode.
int arith_negate_and_encode(uint8_t* out /* [n][32] */, uint32_t n, const uint8_t* pts /* [n][32] */) {
  if (n > 64) return 0;
  fe s[64], den[64], pre[64], prod = fe_one();
  ge_p3 P[64];
  bool zero[64];
  uint32_t w[8];
  for (uint32_t j = 0; j < n; j++) {
    load8(w, pts + 32 * j);
    if (!ristretto_decode(P[j], w)) return 0;
    s[j] = fe_frombytes(w);
    den[j] = negenc_den(s[j], P[j]);
    zero[j] = fe_is_zero(den[j]);
    fe_cmov(den[j], fe_one(), zero[j]);
    pre[j] = prod;
    prod = fe_mul(prod, den[j]);
  }
  fe inv = fe_invert(prod);
  for (uint32_t jj = n; jj > 0; jj--) {
    const uint32_t j = jj - 1;
    const fe inv_j = fe_mul(inv, pre[j]);
    inv = fe_mul(inv, den[j]);
    negenc_finish(w, s[j], P[j], inv_j);
    if (zero[j]) memset(w, 0, sizeof w);
    memcpy(out + 32 * j, w, 32);
  }
  return 1;
}
}  // extern "C"
