// edwards25519 extended-coordinate arithmetic and the ristretto255 group for gfx950, one point per lane.
//
// Replaces curve25519-dalek's RistrettoPoint / CompressedRistretto [3P] at the reference's call sites:
// decompress/compress (/root/reference/src/nizk/presentation.rs:373-412, src/nizk/encryption.rs:172-185,
// src/nizk/issuance.rs:162-189), point +,-,neg (src/nizk/presentation.rs:342-351, encryption.rs:183,185),
// from_uniform_bytes (src/amacs.rs:290).  Formulas: RFC 9496 §4.3 (decode, encode, MAP) and the
// hwcd-2008 a=-1 extended addition/doubling.  Only group elements and canonical encodings are
// contractual: the addition schedules here are this engine's own.
#pragma once
#include "constants.cuh"
#include "fe.cuh"

AFX_DEV fe fe_const(const int32_t* c) {
  fe r;
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) r.v[i] = c[i];
  return r;
}

struct ge_p3 { fe X, Y, Z, T; };        // extended: x = X/Z, y = Y/Z, xy = T/Z
struct ge_p2 { fe X, Y, Z; };           // projective
struct ge_p1p1 { fe X, Y, Z, T; };      // completed: ((X:Z), (Y:T))
struct ge_cached { fe YpX, YmX, Z2, T2d; };   // Y+X, Y-X, 2Z, 2dT: the addition needs 2*Z1*Z2 and gets it in one product
struct ge_niels { fe ypx, ymx, xyd; };         // affine precomputed, HALVED: (y+x)/2, (y-x)/2, d*x*y.  Halving every entry
                                               // scales the completed sum by 1/2 (same point) and saves doubling Z

AFX_DEV ge_p3 ge_identity() {
  ge_p3 r;
  r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero();
  return r;
}
AFX_DEV ge_cached ge_cached_identity() {
  ge_cached r;
  r.YpX = fe_one(); r.YmX = fe_one(); r.Z2 = fe_one(); r.Z2.v[0] = 2; r.T2d = fe_zero();
  return r;
}
AFX_DEV ge_niels ge_niels_identity() {
  ge_niels r;
  r.ypx = fe_const(FEC_INV2); r.ymx = fe_const(FEC_INV2); r.xyd = fe_zero();
  return r;
}
// affine point (x, y) -> halved niels form
AFX_DEV ge_niels ge_niels_from_affine(const fe& x, const fe& y) {
  ge_niels r;
  const fe inv2 = fe_const(FEC_INV2);
  r.ypx = fe_mul(fe_add(y, x), inv2); r.ymx = fe_mul(fe_sub(y, x), inv2); r.xyd = fe_mul(fe_mul(x, y), fe_const(FEC_D));
  return r;
}
// The completed point's bounds (units of fe.cuh: centred 1/2, raw 1; a product needs |f| * |g| <= 3.8, a squaring |f| <= 1.9),
// for every producer in this file:  ge_p2_dbl  |X| < 1.5, |Y| < 1, |Z| < 1, |T| < 2;   ge_add_cached / ge_madd  |X| < 1,
// 0 <= Y < 2, |Z| < 1, 0 <= T < 2.  The four products T*X, Y*Z, T*Z, Y*X therefore stay within 3.
AFX_DEV ge_p2 ge_p1p1_to_p2(const ge_p1p1& p) {
  ge_p2 r;
  r.X = fe_mul(p.T, p.X); r.Y = fe_mul(p.Y, p.Z); r.Z = fe_mul(p.T, p.Z);
  return r;
}
AFX_DEV ge_p3 ge_p1p1_to_p3(const ge_p1p1& p) {
  ge_p3 r;
  r.X = fe_mul(p.T, p.X); r.Y = fe_mul(p.Y, p.Z); r.Z = fe_mul(p.T, p.Z); r.T = fe_mul(p.Y, p.X);
  return r;
}
// Conversions inside a doubling/addition chain, where the consumer of every coordinate is known:
//   GE_FOR_DBL   before a doubling (ge_p2_dbl): X, Y, Z raw - they are squared, and so is Y - X (within 1 unit); T not computed.
//   GE_FOR_ADD   before an addition (ge_add_cached, ge_madd): X, Y, Z, T raw - Y + X (2 units), Y - X, Z and T (1 unit) each meet
//                a table entry of 1 unit.
//   GE_FOR_ANY   everything centred (stores, encodings, table building, any other consumer).
// tests/test_device_arith_on_host.py runs chains of these steps on the host build with every column sum checked
// (AFX_CHECK_BOUNDS).
enum { GE_FOR_DBL = 0, GE_FOR_ADD = 1, GE_FOR_ANY = 3 };
template <int NEXT>
AFX_DEV ge_p3 ge_p1p1_to_p3_for(const ge_p1p1& p) {
  ge_p3 r;
  if constexpr (NEXT == GE_FOR_DBL) {
    r.X = fe_mul_raw(p.T, p.X); r.Y = fe_mul_raw(p.Y, p.Z); r.Z = fe_mul_raw(p.T, p.Z); r.T = r.X;
  } else if constexpr (NEXT == GE_FOR_ADD) {
    r.X = fe_mul_raw(p.T, p.X); r.Y = fe_mul_raw(p.Y, p.Z); r.Z = fe_mul_raw(p.T, p.Z); r.T = fe_mul_raw(p.Y, p.X);
  } else {
    r = ge_p1p1_to_p3(p);
  }
  return r;
}
// the same with a (wave-uniform) run-time consumer
AFX_DEV ge_p3 ge_p1p1_to_p3_next(const ge_p1p1& p, int next) {
  if (next == GE_FOR_DBL) return ge_p1p1_to_p3_for<GE_FOR_DBL>(p);
  if (next == GE_FOR_ADD) return ge_p1p1_to_p3_for<GE_FOR_ADD>(p);
  return ge_p1p1_to_p3(p);
}
AFX_DEV ge_p2 ge_p1p1_to_p2_before_dbl(const ge_p1p1& p) {
  ge_p2 r;
  r.X = fe_mul_raw(p.T, p.X); r.Y = fe_mul_raw(p.Y, p.Z); r.Z = fe_mul_raw(p.T, p.Z);
  return r;
}
AFX_DEV ge_p2 ge_p3_to_p2(const ge_p3& p) {
  ge_p2 r;
  r.X = p.X; r.Y = p.Y; r.Z = p.Z;
  return r;
}
// p: centred coordinates (then Y+X, Y-X and 2Z are within 1 unit, what ge_add_cached expects of an entry)
AFX_DEV ge_cached ge_p3_to_cached(const ge_p3& p) {
  ge_cached r;
  r.YpX = fe_add(p.Y, p.X); r.YmX = fe_sub(p.Y, p.X); r.Z2 = fe_add(p.Z, p.Z); r.T2d = fe_mul(p.T, fe_const(FEC_D2));
  return r;
}
// cached form with (YpX, YmX) carried so the entry can itself be added to lazily (tables in memory)
AFX_DEV ge_cached ge_p3_to_cached_reduced(const ge_p3& p) {
  ge_cached r = ge_p3_to_cached(p);
  r.YpX = fe_carry(r.YpX); r.YmX = fe_carry(r.YmX);
  return r;
}
AFX_DEV ge_p1p1 ge_p2_dbl(const ge_p2& p) {
  ge_p1p1 r;
  // 2XY is taken from (Y - X)^2 = XX + YY - 2XY rather than from (X + Y)^2: the difference of two raw coordinates is
  // within 1 unit, so X and Y may both be raw when they come here (their sum would be no valid squaring input).
  // XX, YY, ZZ raw (their rounding would sit on the squarings' serial path); the two sums of raw values, YY + XX and 2 ZZ
  // (limbs in [0, 2)), are brought to (-1, 1) by subtracting p limb by limb, which costs nine additions off that path;
  // only (Y-X)^2 is centred.  Then |Y3| < 1, |Z3| < 1, |X3| = |Y3 - AA| < 1.5, |T3| = |2ZZ - p - Z3| < 2.
  fe XX = fe_sq_raw(p.X), YY = fe_sq_raw(p.Y), ZZ = fe_sq_raw(p.Z);
  fe B = fe_sub_p(fe_add(ZZ, ZZ));
  fe A = fe_sub(p.Y, p.X);
  fe AA = fe_sq(A);
  r.Y = fe_sub_p(fe_add(YY, XX));
  r.Z = fe_sub(YY, XX);
  r.X = fe_sub(r.Y, AA);
  r.T = fe_sub(B, r.Z);
  return r;
}
// p + q (q cached).  neg: subtract instead (swap the two products' partners, flip the sign of C).
AFX_DEV ge_p1p1 ge_add_cached(const ge_p3& p, const ge_cached& q, bool neg) {
  ge_p1p1 r;
  fe qp = q.YpX, qm = q.YmX;
  fe_cswap(qp, qm, neg);
  // A, B raw: they only meet in X3 = A - B (within 1 unit) and Y3 = A + B (2 units)
  fe A = fe_mul_raw(fe_add(p.Y, p.X), qp);
  fe B = fe_mul_raw(fe_sub(p.Y, p.X), qm);
  // C' = -(2dT2) T1 (the sign flipped again when subtracting), raw: Z3 = D + C = D - C' is then a difference of raw
  // values (within 1 unit) and T3 = D - C = D + C' a sum (2 units)
  fe Cn = fe_mul_raw(fe_cneg(q.T2d, !neg), p.T);
  fe D = fe_mul_raw(p.Z, q.Z2);   // 2 Z1 Z2
  r.X = fe_sub(A, B);
  r.Y = fe_add(A, B);
  r.Z = fe_sub(D, Cn);
  r.T = fe_add(D, Cn);
  return r;
}
AFX_DEV ge_p1p1 ge_madd(const ge_p3& p, const ge_niels& q, bool neg) {
  ge_p1p1 r;
  fe qp = q.ypx, qm = q.ymx;
  fe_cswap(qp, qm, neg);
  fe A = fe_mul_raw(fe_add(p.Y, p.X), qp);   // raw: see ge_add_cached
  fe B = fe_mul_raw(fe_sub(p.Y, p.X), qm);
  fe Cn = fe_mul_raw(fe_cneg(q.xyd, !neg), p.T);   // -C, raw: see ge_add_cached
  const fe& D = p.Z;   // entries are halved: no doubling of Z
  r.X = fe_sub(A, B);
  r.Y = fe_add(A, B);
  r.Z = fe_sub(D, Cn);
  r.T = fe_add(D, Cn);
  return r;
}
AFX_DEV ge_p3 ge_add(const ge_p3& p, const ge_p3& q) { return ge_p1p1_to_p3(ge_add_cached(p, ge_p3_to_cached(q), false)); }
AFX_DEV ge_p3 ge_sub(const ge_p3& p, const ge_p3& q) { return ge_p1p1_to_p3(ge_add_cached(p, ge_p3_to_cached(q), true)); }
AFX_DEV ge_p3 ge_neg(const ge_p3& p) {
  ge_p3 r;
  r.X = fe_neg(p.X); r.Y = p.Y; r.Z = p.Z; r.T = fe_neg(p.T);
  return r;
}
AFX_DEV ge_p3 ge_double(const ge_p3& p) { return ge_p1p1_to_p3(ge_p2_dbl(ge_p3_to_p2(p))); }
AFX_DEV ge_p3 ge_carry(const ge_p3& p) {
  ge_p3 r;
  r.X = fe_carry(p.X); r.Y = fe_carry(p.Y); r.Z = fe_carry(p.Z); r.T = fe_carry(p.T);
  return r;
}

// RFC 9496 §4.2 SQRT_RATIO_M1(u, v) (dalek FieldElement::sqrt_ratio_i).  u, v reduced or one add deep.
AFX_DEV bool fe_sqrt_ratio_i(fe& r_out, const fe& u, const fe& v) {
  const fe sqrt_m1 = fe_const(FEC_SQRT_M1);
  fe v3 = fe_mul(fe_sq(v), v);
  fe v7 = fe_mul(fe_sq(v3), v);
  fe r = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
  fe check = fe_mul(v, fe_sq(r));
  fe uc = fe_carry(u);
  fe neg_u = fe_neg(uc);
  fe neg_u_i = fe_mul(neg_u, sqrt_m1);
  const bool correct = fe_eq(check, uc);
  const bool flipped = fe_eq(check, neg_u);
  const bool flipped_i = fe_eq(check, neg_u_i);
  fe r_prime = fe_mul(r, sqrt_m1);
  fe_cmov(r, r_prime, flipped | flipped_i);
  r_out = fe_abs(r);
  return correct | flipped;
}

// RFC 9496 §4.3.1 Decode (CompressedRistretto::decompress).  w = 8 LE dwords.  Returns false on any
// rejection; on success r is the extended point with Z = 1 and all limbs reduced.
AFX_DEV bool ristretto_decode(ge_p3& r, const uint32_t w[8]) {
  fe s = fe_frombytes(w);
  uint32_t chk[8];
  fe_tobytes(chk, s);
  bool canonical = true;
#pragma unroll
  for (int i = 0; i < 8; i++) canonical &= (chk[i] == w[i]);
  const bool s_neg = (chk[0] & 1) != 0;
  const fe one = fe_one();
  fe ss = fe_sq(s);
  fe u1 = fe_sub(one, ss);
  fe u2 = fe_add(one, ss);
  fe u2s = fe_sq(u2);
  fe v = fe_carry(fe_sub(fe_neg(fe_mul(fe_const(FEC_D), fe_sq(u1))), u2s));
  fe I;
  const bool was_square = fe_sqrt_ratio_i(I, one, fe_mul(v, u2s));
  fe Dx = fe_mul(I, u2);
  fe Dy = fe_mul(fe_mul(I, Dx), v);
  fe sDx = fe_mul(s, Dx);
  fe x = fe_abs(fe_add(sDx, sDx));
  fe y = fe_mul(u1, Dy);
  fe t = fe_mul(x, y);
  const bool ok = canonical & !s_neg & was_square & !fe_is_negative(t) & !fe_is_zero(y);
  r.X = fe_carry(x); r.Y = y; r.Z = one; r.T = t;
  return ok;
}

// RFC 9496 §4.3.2 Encode (RistrettoPoint::compress).  p: limbs reduced (outputs of fe_mul / ge_carry).
// the part of Encode after the inverse square root: u1 = (Z+Y)(Z-Y), u2 = XY, I = 1/sqrt(u1 u2^2) (the non-negative root)
AFX_DEV void ristretto_encode_tail(uint32_t w[8], const ge_p3& p, const fe& u1, const fe& u2, const fe& I);
AFX_DEV void ristretto_encode(uint32_t w[8], const ge_p3& p) {
  const fe one = fe_one();
  fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
  fe u2 = fe_mul(p.X, p.Y);
  fe I;
  fe_sqrt_ratio_i(I, one, fe_mul(u1, fe_sq(u2)));
  ristretto_encode_tail(w, p, u1, u2, I);
}
AFX_DEV void ristretto_encode_tail(uint32_t w[8], const ge_p3& p, const fe& u1, const fe& u2, const fe& I) {
  fe D1 = fe_mul(u1, I);
  fe D2 = fe_mul(u2, I);
  fe Zinv = fe_mul(fe_mul(D1, D2), p.T);
  const fe sqrt_m1 = fe_const(FEC_SQRT_M1);
  fe ix = fe_mul(p.X, sqrt_m1);
  fe iy = fe_mul(p.Y, sqrt_m1);
  fe ead = fe_mul(D1, fe_const(FEC_INVSQRT_A_MINUS_D));
  const bool rotate = fe_is_negative(fe_mul(p.T, Zinv));
  fe x = p.X, y = p.Y, Dinv = D2;
  fe_cmov(x, iy, rotate);
  fe_cmov(y, ix, rotate);
  fe_cmov(Dinv, ead, rotate);
  y = fe_cneg(y, fe_is_negative(fe_mul(x, Zinv)));
  fe s = fe_abs(fe_mul(Dinv, fe_sub(p.Z, y)));
  fe_tobytes(w, s);
}

// The encoding of the NEGATION of a decoded point without a square root.  Decode(s) = (x, y) with y = u1/u2, u1 = 1 - s^2,
// u2 = 1 + s^2, and x^2 = 4 s^2 / v for the v whose inverse square root Decode took.  Encode(-x, y) needs
// 1/sqrt((1 - y^2) x^2 y^2), and (1 - y^2) x^2 y^2 = (2 x s u1 / u2^2)^2 - a perfect square of quantities at hand: the root is
// +-u2^2 / (2 x s u1), ONE INVERSION, which a caller with several such points shares among them (Montgomery's trick,
// k_negenc).  neg_den() is the value to invert; neg_finish() the encoding, given its inverse.  x = 0 or s = 0 (the identity,
// or the placeholder of a failed decode) gives den = 0: the caller leaves such a factor out and the result is not used.
AFX_DEV fe negenc_den(const fe& s, const ge_p3& P) {
  const fe u1 = fe_sub(fe_one(), fe_sq(s));
  return fe_mul(fe_mul(fe_add(P.X, P.X), s), u1);
}
AFX_DEV void negenc_finish(uint32_t w[8], const fe& s, const ge_p3& P, const fe& inv_den) {
  fe u2d = fe_add(fe_one(), fe_sq(s));                       // 1 + s^2
  const fe I = fe_abs(fe_mul(fe_sq(u2d), inv_den));          // the non-negative root, as SQRT_RATIO_M1 returns it
  const ge_p3 N = ge_neg(P);                                 // (-x, y, 1, -xy)
  const fe u1 = fe_mul(fe_add(N.Z, N.Y), fe_sub(N.Z, N.Y));
  const fe u2 = fe_mul(N.X, N.Y);
  ristretto_encode_tail(w, N, u1, u2, I);
}

// The encoding of TWICE a point without a square root (curve25519-dalek's double_and_compress_batch [3P]; kernels.hip
// k_compress2x has the story): with e = 2XY, f = Z^2 + dT^2, g = Y^2 + X^2, h = Z^2 - dT^2, 2P = (e*f : g*h : f*g : e*h), and the
// encoding needs only 1/(e*g) and 1/(f*h) - that is 1/(e*f*g*h), an inversion that can be shared among many points.
// p: limbs reduced.  zero: e*f*g*h = 0, true exactly for the representatives of the identity; efgh is then set to 1 so that
// the caller's running product is not spoilt.
struct c2x_state { fe e, f, g, h, eg, fh, efgh; bool zero; };
AFX_DEV c2x_state c2x_from(const ge_p3& P) {
  c2x_state s;
  const fe XX = fe_sq(P.X), YY = fe_sq(P.Y), ZZ = fe_sq(P.Z), dTT = fe_mul(fe_sq(P.T), fe_const(FEC_D));
  s.e = fe_mul(fe_add(P.Y, P.Y), P.X);
  s.f = fe_add(ZZ, dTT);
  s.g = fe_add(YY, XX);
  s.h = fe_sub(ZZ, dTT);
  s.eg = fe_mul(s.g, s.e);
  s.fh = fe_mul(s.f, s.h);
  s.efgh = fe_mul(s.eg, s.fh);
  s.zero = fe_is_zero(s.efgh);
  fe_cmov(s.efgh, fe_one(), s.zero);
  return s;
}
// w = encoding of 2P given inv = 1 / (e*f*g*h) of the same state (all zeros for the identity)
AFX_DEV void c2x_finish(uint32_t w[8], const c2x_state& s, const fe& inv) {
  const fe Zinv = fe_mul(s.eg, inv), Tinv = fe_mul(s.fh, inv);
  const bool rotate = fe_is_negative(fe_mul(s.eg, Zinv));
  fe e = s.e, g = s.g, h = s.h, magic = fe_const(FEC_INVSQRT_A_MINUS_D);
  fe_cmov(e, s.g, rotate);
  fe_cmov(g, fe_neg(s.e), rotate);
  fe_cmov(h, fe_mul(s.f, fe_const(FEC_SQRT_M1)), rotate);
  fe_cmov(magic, fe_const(FEC_SQRT_M1), rotate);
  g = fe_cneg(g, fe_is_negative(fe_mul(fe_mul(h, e), Zinv)));
  const fe sres = fe_abs(fe_mul(fe_sub(h, g), fe_mul(magic, fe_mul(g, Tinv))));
  fe_tobytes(w, sres);
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = s.zero ? 0u : w[i];
}

// RFC 9496 §4.3.4 MAP
AFX_DEV ge_p3 ristretto_elligator(const fe& r0) {
  const fe one = fe_one();
  const fe d = fe_const(FEC_D);
  fe r = fe_mul(fe_const(FEC_SQRT_M1), fe_sq(r0));
  fe u = fe_mul(fe_add(r, one), fe_const(FEC_ONE_MINUS_D_SQ));
  fe c = fe_neg(one);
  fe v = fe_mul(fe_sub(c, fe_mul(r, d)), fe_add(r, d));
  fe s;
  const bool was_square = fe_sqrt_ratio_i(s, u, v);
  fe s_prime = fe_neg(fe_abs(fe_mul(s, r0)));
  fe_cmov(s, s_prime, !was_square);
  fe_cmov(c, r, !was_square);
  fe N = fe_carry(fe_sub(fe_mul(fe_mul(c, fe_sub(r, one)), fe_const(FEC_D_MINUS_ONE_SQ)), v));
  fe w0 = fe_mul(fe_add(s, s), v);
  fe w1 = fe_mul(N, fe_const(FEC_SQRT_AD_MINUS_ONE));
  fe ss = fe_sq(s);
  fe w2 = fe_sub(one, ss);
  fe w3 = fe_add(one, ss);
  ge_p3 p;
  p.X = fe_mul(w0, w3); p.Y = fe_mul(w2, w1); p.Z = fe_mul(w1, w3); p.T = fe_mul(w0, w2);
  return p;
}
// RistrettoPoint::from_uniform_bytes: w = 16 LE dwords
AFX_DEV ge_p3 ristretto_from_uniform(const uint32_t w[16]) {
  ge_p3 p1 = ristretto_elligator(fe_frombytes(w));
  ge_p3 p2 = ristretto_elligator(fe_frombytes(w + 8));
  return ge_add(p1, p2);
}
AFX_DEV bool is_identity_encoding(const uint32_t w[8]) {
  return (w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0;
}
