// Scalars mod l = 2^252 + 27742317777372353535851937790883648493 on gfx950: 8 x 32-bit limbs per lane.
// Replaces curve25519-dalek `Scalar` [3P] at the reference's call sites: from_bytes_mod_order_wide
// (challenge derivation inside zkp; /root/reference/src/amacs.rs:289), s*c + b responses (zkp prove_compact),
// -t*z and -z(a0+a1*m3) (src/nizk/presentation.rs:163, src/nizk/encryption.rs:78), canonicity
// (src/amacs.rs:141-149).  Reduction folds at bit 252 with 2^252 = -delta (mod l).
#pragma once
#include "fe.cuh"

struct sc {
  uint32_t v[8];
};

__device__ __constant__ const uint32_t SC_DELTA[4] = { 0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu };
__device__ __constant__ const uint32_t SC_L[8] = { 0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u };

template <int NA, int NB>
AFX_DEV void mp_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
#pragma unroll
  for (int i = 0; i < NA + NB; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      c += (uint64_t)a[i] * b[j] + r[i + j];
      r[i + j] = (uint32_t)c;
      c >>= 32;
    }
    r[i + NB] = (uint32_t)c;
  }
}

// lo = x mod 2^252 (8 limbs), hi = x >> 252 (NHI limbs) for an N-limb x
template <int N, int NHI>
AFX_DEV void mp_split252(uint32_t lo[8], uint32_t* hi, const uint32_t* x) {
#pragma unroll
  for (int i = 0; i < 8; i++) lo[i] = i < N ? x[i] : 0;
  lo[7] &= 0x0fffffffu;
#pragma unroll
  for (int i = 0; i < NHI; i++) {
    const uint32_t a = (i + 7 < N) ? x[i + 7] : 0, b = (i + 8 < N) ? x[i + 8] : 0;
    hi[i] = (a >> 28) | (b << 4);
  }
}

// 9-limb signed-safe add/sub helpers (values stay non-negative by construction)
AFX_DEV void mp9_add8(uint32_t acc[9], const uint32_t b[8]) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)acc[i] + b[i]; acc[i] = (uint32_t)c; c >>= 32; }
  acc[8] += (uint32_t)c;
}
AFX_DEV uint32_t mp9_sub8(uint32_t out[9], const uint32_t acc[9], const uint32_t b[8]) {
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint64_t bi = (i < 8 ? (uint64_t)b[i] : 0) + borrow;
    const uint64_t t = (uint64_t)acc[i] - bi;
    out[i] = (uint32_t)t;
    borrow = (uint32_t)(t >> 63);
  }
  return borrow;
}

// x (16 limbs, 512 bits) mod l
AFX_DEV sc sc_reduce512(const uint32_t x[16]) {
  uint32_t lo1[8], hi1[9], t1[13], lo2[8], hi2[5], t2[9], lo3[8], hi3[1], t3[5];
  uint32_t delta[4], ell[8];
#pragma unroll
  for (int i = 0; i < 4; i++) delta[i] = SC_DELTA[i];
#pragma unroll
  for (int i = 0; i < 8; i++) ell[i] = SC_L[i];
  mp_split252<16, 9>(lo1, hi1, x);       // hi1 < 2^260
  mp_mul<9, 4>(t1, hi1, delta);          // < 2^385
  mp_split252<13, 5>(lo2, hi2, t1);      // hi2 < 2^133
  mp_mul<5, 4>(t2, hi2, delta);          // < 2^258
  mp_split252<9, 1>(lo3, hi3, t2);       // hi3 < 2^6
  mp_mul<1, 4>(t3, hi3, delta);          // < 2^131
  // x = lo1 - lo2 + lo3 - t3 (mod l); add 2l first so the running value stays >= 0
  uint32_t acc[9], tmp[9], t3w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = lo1[i];
  acc[8] = 0;
  mp9_add8(acc, lo3);
  mp9_add8(acc, ell);
  mp9_add8(acc, ell);
  mp9_sub8(acc, acc, lo2);
#pragma unroll
  for (int i = 0; i < 8; i++) t3w[i] = i < 5 ? t3[i] : 0;
  mp9_sub8(acc, acc, t3w);
  // acc < 4l: subtract l while it fits
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t borrow = mp9_sub8(tmp, acc, ell);
#pragma unroll
    for (int i = 0; i < 9; i++) acc[i] = borrow ? acc[i] : tmp[i];
  }
  sc r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = acc[i];
  return r;
}

AFX_DEV bool sc_is_canonical(const sc& a) {
  // a < l  <=>  a - l borrows
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)a.v[i] - SC_L[i] - borrow;
    borrow = (uint32_t)(t >> 63);
  }
  return borrow != 0;
}
AFX_DEV bool sc_eq(const sc& a, const sc& b) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d |= a.v[i] ^ b.v[i];
  return d == 0;
}
AFX_DEV sc sc_mul(const sc& a, const sc& b) {
  uint32_t p[16];
  mp_mul<8, 8>(p, a.v, b.v);
  return sc_reduce512(p);
}
// a*b + c mod l
AFX_DEV sc sc_muladd(const sc& a, const sc& b, const sc& c) {
  uint32_t p[16];
  mp_mul<8, 8>(p, a.v, b.v);
  uint64_t cy = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    cy += (uint64_t)p[i] + (i < 8 ? c.v[i] : 0);
    p[i] = (uint32_t)cy;
    cy >>= 32;
  }
  return sc_reduce512(p);
}
// -a mod l for canonical a
AFX_DEV sc sc_neg(const sc& a) {
  uint32_t x[16];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)SC_L[i] - a.v[i] - borrow;
    x[i] = (uint32_t)t;
    borrow = (uint32_t)(t >> 63);
  }
#pragma unroll
  for (int i = 8; i < 16; i++) x[i] = 0;
  return sc_reduce512(x);  // maps l (a == 0) back to 0
}
AFX_DEV sc sc_load(const uint8_t* p) {  // p 4-byte aligned
  sc r;
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = q[i];
  return r;
}
AFX_DEV void sc_store(uint8_t* p, const sc& a) {
  uint32_t* q = reinterpret_cast<uint32_t*>(p);
#pragma unroll
  for (int i = 0; i < 8; i++) q[i] = a.v[i];
}
// Signed fixed-window recoding without carries: with s' = s + 0x88..88, digit_i = nibble_i(s') - 8 in [-8, 7]
// and sum digit_i 16^i = s; with s' = s + 0x80..80, digit_i = byte_i(s') - 128 in [-128, 127] and
// sum digit_i 256^i = s.  Needs s < 2^255 (any canonical scalar).
// The same for windows of B bits that do not divide 32: bias = sum over windows of 2^(B-1) * 2^(B*j); the result
// has up to 253 + B bits (9 words).  digit_j = ((s' >> B*j) & (2^B - 1)) - 2^(B-1).
template <int B, int WINDOWS>
AFX_DEV void sc_bias_wide(uint32_t out[9], const sc& s) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint32_t bias = 0;
#pragma unroll
    for (int j = 0; j < WINDOWS; j++) {
      const int bit = B * j + B - 1;
      if (bit / 32 == i) bias |= 1u << (bit % 32);
    }
    c += (uint64_t)(i < 8 ? s.v[i] : 0u) + bias;
    out[i] = (uint32_t)c;
    c >>= 32;
  }
}
// s * 2^-1 mod l for canonical s: (s + l) / 2 when s is odd, s / 2 otherwise.  (A non-canonical s - its item has failed already -
// just gives some 256-bit value.)
AFX_DEV sc sc_half(const sc& s) {
  const uint32_t odd = s.v[0] & 1u;
  uint32_t t[9];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)s.v[i] + (odd ? SC_L[i] : 0u);
    t[i] = (uint32_t)c;
    c >>= 32;
  }
  t[8] = (uint32_t)c;
  sc r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
  return r;
}
// 2 s mod l for canonical s (2 s < 2 l: one conditional subtraction)
AFX_DEV sc sc_dbl(const sc& s) {
  uint32_t t[9], u[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = (i < 8 ? s.v[i] << 1 : 0u) | (i > 0 ? s.v[i - 1] >> 31 : 0u);
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint64_t d = (uint64_t)t[i] - (i < 8 ? SC_L[i] : 0u) - borrow;
    u[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
  sc r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = borrow ? t[i] : u[i];
  return r;
}
AFX_DEV void sc_bias(uint32_t out[8], const sc& s, uint32_t bias) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)s.v[i] + bias;
    out[i] = (uint32_t)c;
    c >>= 32;
  }
}
