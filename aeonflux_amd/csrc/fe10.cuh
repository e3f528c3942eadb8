// The inversion and square-root chains (z^(p-2), z^((p-5)/8): 254 / 251 squarings and 11 products in a row) run in the
// 10 x 25.5-bit form of GF(2^255-19) - limbs of 26 and 25 bits alternating, the form the whole engine used before round 2.
// Its squaring needs no fold of high columns (a 25.5-bit limb times 19 or 38 still fits int32, so the wrapped terms are
// premultiplied 32-bit operands): 55 multiply-adds + 16 32-bit preparations against the 9 x 29-bit form's 45 + 16 folds + 1 -
// 5 % fewer issue cycles per squaring (profiles/r02_fe9_rates.txt), while its product is 3 % slower (100 multiply-adds against
// 98).  A chain is 96 % squarings, everything else in the engine is mostly products: so the chains alone take this form, entered
// and left through the canonical 32-byte encoding (fe_tobytes / fe_frombytes: ~1 % of a chain).
// Bounds (checked on the actual operands by the host build, AFX_CHECK_BOUNDS): a raw result has limbs in [0, 2^26) / [0, 2^25)
// (+ a carry on limb 1); it is valid as either operand of a product and as the input of a squaring - all a chain does.
#pragma once

struct fe10 {
  int32_t v[10];
};
#ifdef AFX_CHECK_BOUNDS
static inline void afx_check_products10(const int32_t* f, const int32_t* g, bool square) {
  for (int i = 0; i < 10; i++) {
    const int64_t ag = g[i] < 0 ? -(int64_t)g[i] : g[i], af = f[i] < 0 ? -(int64_t)f[i] : f[i];
    // limb 0 is never a wrapped term's second factor: its 19-fold is computed but not consumed
    if (i != 0 && ag * (square && (i & 1) ? 38 : 19) >= (1LL << 31)) afx_bounds_violation("19x/38x premultiplication overflows int32");
    if (af * 2 >= (1LL << 31)) afx_bounds_violation("2x premultiplication overflows int32");
  }
  for (int k = 0; k < 10; k++) {
    unsigned __int128 sum = 0;
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      const unsigned __int128 af = f[i] < 0 ? -(int64_t)f[i] : f[i], ag = g[j] < 0 ? -(int64_t)g[j] : g[j];
      sum += af * ag * (((i & 1) && (j & 1)) ? 2 : 1) * (i > k ? 19 : 1);
    }
    if (sum >= ((unsigned __int128)1 << 62)) afx_bounds_violation("column sum beyond 2^62");
  }
}
#define AFX_CHECK_MUL10(f, g) afx_check_products10((f).v, (g).v, false)
#define AFX_CHECK_SQ10(f) afx_check_products10((f).v, (f).v, true)
#else
#define AFX_CHECK_MUL10(f, g) ((void)0)
#define AFX_CHECK_SQ10(f) ((void)0)
#endif
#ifdef AFX_COUNT_OPS
extern thread_local uint64_t afx_n_chain_mul, afx_n_chain_sq;   // the share of afx_n_mul / afx_n_sq executed here
#endif

// carry chain over 64-bit limb values -> centred limbs
AFX_DEV fe10 fe10_carry64(int64_t h[10]) {
  int64_t c;
  c = (h[0] + (1LL << 25)) >> 26; h[1] += c; h[0] -= c << 26;
  c = (h[4] + (1LL << 25)) >> 26; h[5] += c; h[4] -= c << 26;
  c = (h[1] + (1LL << 24)) >> 25; h[2] += c; h[1] -= c << 25;
  c = (h[5] + (1LL << 24)) >> 25; h[6] += c; h[5] -= c << 25;
  c = (h[2] + (1LL << 25)) >> 26; h[3] += c; h[2] -= c << 26;
  c = (h[6] + (1LL << 25)) >> 26; h[7] += c; h[6] -= c << 26;
  c = (h[3] + (1LL << 24)) >> 25; h[4] += c; h[3] -= c << 25;
  c = (h[7] + (1LL << 24)) >> 25; h[8] += c; h[7] -= c << 25;
  c = (h[4] + (1LL << 25)) >> 26; h[5] += c; h[4] -= c << 26;
  c = (h[8] + (1LL << 25)) >> 26; h[9] += c; h[8] -= c << 26;
  c = (h[9] + (1LL << 24)) >> 25; h[0] += c * 19; h[9] -= c << 25;
  c = (h[0] + (1LL << 25)) >> 26; h[1] += c; h[0] -= c << 26;
  fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = (int32_t)h[i];
  return r;
}

AFX_DEV fe10 fe10_carry(const fe10& f) {
  int64_t h[10];
#pragma unroll
  for (int i = 0; i < 10; i++) h[i] = f.v[i];
  return fe10_carry64(h);
}

// Schoolbook product, columns in order 0..9: column k's mad chain starts from the carry out of column k-1 (the mad's 64-bit
// addend).  CMASK: bit k set = limb k comes out centred (its rounding constant travels in the carry of column k-1), clear = raw.
template <uint32_t CMASK>
AFX_DEV fe10 fe10_mul_impl(const fe10& f, const fe10& g) {
  AFX_COUNT(afx_n_mul); AFX_COUNT(afx_n_chain_mul);
  AFX_CHECK_MUL10(f, g);
  int32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    g19[i] = (int32_t)(19u * (uint32_t)g.v[i]);
    f2[i] = (int32_t)(2u * (uint32_t)f.v[i]);
  }
  fe10 r;
  int64_t c = (CMASK & 1u) ? (1LL << 25) : 0;  // rounding constant of limb 0; later carries arrive with the next limb's folded in
  uint32_t u0 = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    int64_t H = c;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      const bool wrap = i > k;
      const int32_t a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
      const int32_t b = wrap ? g19[j] : g.v[j];
      H += (int64_t)a * (int64_t)b;
      if (k > 0 || i > 0) AFX_PIN(H);   // (column 0 starts from a constant: its first product, pinned, would be computed twice - fe.cuh fe_mul_impl)
    }
    const int bits = (k & 1) ? 25 : 26;
    const uint32_t lo = (uint32_t)H & ((1u << bits) - 1);
    if (k == 0) u0 = lo; else r.v[k] = ((CMASK >> k) & 1u) ? (int32_t)lo - (1 << (bits - 1)) : (int32_t)lo;
    c = (k < 9 && ((CMASK >> (k + 1)) & 1u)) ? ((H + (1LL << 50)) >> bits) : (H >> bits);
  }
  // wrap: limb 0 gets 19 * carry(limb 9); u0 still holds limb 0 (with its rounding constant when centred)
  int64_t H0 = (int64_t)u0 + c * 19;
  const int32_t c0 = (int32_t)(H0 >> 26);
  r.v[0] = (CMASK & 1u) ? (int32_t)((uint32_t)H0 & 0x3ffffffu) - (1 << 25) : (int32_t)((uint32_t)H0 & 0x3ffffffu);
  r.v[1] += c0;
  return r;
}
AFX_DEV fe10 fe10_mul_raw(const fe10& f, const fe10& g) { return fe10_mul_impl<0u>(f, g); }

template <uint32_t CMASK>
AFX_DEV fe10 fe10_sq_impl(const fe10& f) {
  AFX_COUNT(afx_n_sq); AFX_COUNT(afx_n_chain_sq);
  AFX_CHECK_SQ10(f);
  int32_t f2[10], f19[10], f38[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    f2[i] = (int32_t)(2u * (uint32_t)f.v[i]);
    f19[i] = (int32_t)(19u * (uint32_t)f.v[i]);
    f38[i] = (int32_t)(38u * (uint32_t)f.v[i]);
  }
  fe10 r;
  int64_t c = (CMASK & 1u) ? (1LL << 25) : 0;
  uint32_t u0 = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    int64_t H = c;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      if (j < i) continue;
      const bool wrap = i + j >= 10;
      const bool odd2 = (i & 1) && (j & 1);
      const int32_t a = (i == j) ? f.v[i] : f2[i];
      const int32_t b = wrap ? (odd2 ? f38[j] : f19[j]) : (odd2 ? f2[j] : f.v[j]);
      H += (int64_t)a * (int64_t)b;
      if (k > 0 || i > 0) AFX_PIN(H);   // (column 0 starts from a constant: its first product, pinned, would be computed twice - fe.cuh fe_mul_impl)
    }
    const int bits = (k & 1) ? 25 : 26;
    const uint32_t lo = (uint32_t)H & ((1u << bits) - 1);
    if (k == 0) u0 = lo; else r.v[k] = ((CMASK >> k) & 1u) ? (int32_t)lo - (1 << (bits - 1)) : (int32_t)lo;
    c = (k < 9 && ((CMASK >> (k + 1)) & 1u)) ? ((H + (1LL << 50)) >> bits) : (H >> bits);
  }
  int64_t H0 = (int64_t)u0 + c * 19;
  const int32_t c0 = (int32_t)(H0 >> 26);
  r.v[0] = (CMASK & 1u) ? (int32_t)((uint32_t)H0 & 0x3ffffffu) - (1 << 25) : (int32_t)((uint32_t)H0 & 0x3ffffffu);
  r.v[1] += c0;
  return r;
}
AFX_DEV fe10 fe10_sq_raw(const fe10& f) { return fe10_sq_impl<0u>(f); }
// f^(2^n), n >= 1, rolled loop
AFX_DEV fe10 fe10_sqn(fe10 f, int n) {
#pragma unroll 1
  for (int i = 0; i < n; i++) f = fe10_sq_raw(f);
  return f;
}

// from 8 little-endian dwords (bit 255 ignored): unsigned limbs < 2^26 / 2^25
AFX_DEV fe10 fe10_frombytes(const uint32_t w[8]) {
  fe10 r;
  const uint64_t w01 = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
  const uint64_t w12 = (uint64_t)w[1] | ((uint64_t)w[2] << 32);
  const uint64_t w23 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
  const uint64_t w34 = (uint64_t)w[3] | ((uint64_t)w[4] << 32);
  const uint64_t w45 = (uint64_t)w[4] | ((uint64_t)w[5] << 32);
  const uint64_t w56 = (uint64_t)w[5] | ((uint64_t)w[6] << 32);
  const uint64_t w67 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
  r.v[0] = (int32_t)(w01 & 0x3ffffff);                 // bits   0.. 25
  r.v[1] = (int32_t)((w01 >> 26) & 0x1ffffff);         // bits  26.. 50
  r.v[2] = (int32_t)((w12 >> 19) & 0x3ffffff);         // bits  51.. 76   (51-32 = 19)
  r.v[3] = (int32_t)((w23 >> 13) & 0x1ffffff);         // bits  77..101   (77-64 = 13)
  r.v[4] = (int32_t)((w34 >> 6) & 0x3ffffff);          // bits 102..127   (102-96 = 6)
  r.v[5] = (int32_t)(w45 & 0x1ffffff);                 // bits 128..152
  r.v[6] = (int32_t)((w45 >> 25) & 0x3ffffff);         // bits 153..178
  r.v[7] = (int32_t)((w56 >> 19) & 0x1ffffff);         // bits 179..203   (179-160 = 19)
  r.v[8] = (int32_t)((w67 >> 12) & 0x3ffffff);         // bits 204..229   (204-192 = 12)
  r.v[9] = (int32_t)((w[7] >> 6) & 0x1ffffff);         // bits 230..254   (230-224 = 6), bit 255 dropped
  return r;
}


// canonical little-endian encoding into 8 dwords; input: any limbs fe10_carry accepts
AFX_DEV void fe10_tobytes(uint32_t w[8], const fe10& f) {
  fe10 t = fe10_carry(f);
  int32_t h0 = t.v[0], h1 = t.v[1], h2 = t.v[2], h3 = t.v[3], h4 = t.v[4];
  int32_t h5 = t.v[5], h6 = t.v[6], h7 = t.v[7], h8 = t.v[8], h9 = t.v[9];
  int32_t q = (19 * h9 + (1 << 24)) >> 25;
  q = (h0 + q) >> 26; q = (h1 + q) >> 25; q = (h2 + q) >> 26; q = (h3 + q) >> 25; q = (h4 + q) >> 26;
  q = (h5 + q) >> 25; q = (h6 + q) >> 26; q = (h7 + q) >> 25; q = (h8 + q) >> 26; q = (h9 + q) >> 25;
  h0 += 19 * q;
  int32_t c;
  c = h0 >> 26; h1 += c; h0 -= c << 26;
  c = h1 >> 25; h2 += c; h1 -= c << 25;
  c = h2 >> 26; h3 += c; h2 -= c << 26;
  c = h3 >> 25; h4 += c; h3 -= c << 25;
  c = h4 >> 26; h5 += c; h4 -= c << 26;
  c = h5 >> 25; h6 += c; h5 -= c << 25;
  c = h6 >> 26; h7 += c; h6 -= c << 26;
  c = h7 >> 25; h8 += c; h7 -= c << 25;
  c = h8 >> 26; h9 += c; h8 -= c << 26;
  c = h9 >> 25; h9 -= c << 25;
  // all limbs now in [0, 2^26) / [0, 2^25): pack at bit offsets 0,26,51,77,102,128,153,179,204,230
  const uint64_t a = (uint64_t)(uint32_t)h0 | ((uint64_t)(uint32_t)h1 << 26) | ((uint64_t)(uint32_t)h2 << 51);  // bits 0..76 (overflowing part dropped)
  w[0] = (uint32_t)a;
  w[1] = (uint32_t)(a >> 32);
  const uint64_t b = ((uint64_t)(uint32_t)h2 >> 13) | ((uint64_t)(uint32_t)h3 << 13) | ((uint64_t)(uint32_t)h4 << 38);  // bits 64..127
  w[2] = (uint32_t)b;
  w[3] = (uint32_t)(b >> 32);
  const uint64_t d = (uint64_t)(uint32_t)h5 | ((uint64_t)(uint32_t)h6 << 25) | ((uint64_t)(uint32_t)h7 << 51);  // bits 128..191
  w[4] = (uint32_t)d;
  w[5] = (uint32_t)(d >> 32);
  const uint64_t e = ((uint64_t)(uint32_t)h7 >> 13) | ((uint64_t)(uint32_t)h8 << 12) | ((uint64_t)(uint32_t)h9 << 38);  // bits 192..255
  w[6] = (uint32_t)e;
  w[7] = (uint32_t)(e >> 32);
}


// z^(2^250-1) and z^11
AFX_DEV void fe10_pow22501(fe10& t250, fe10& z11, const fe10& z) {
  fe10 z2 = fe10_sq_raw(z);
  fe10 z8 = fe10_sqn(z2, 2);
  fe10 z9 = fe10_mul_raw(z, z8);
  z11 = fe10_mul_raw(z2, z9);
  fe10 z22 = fe10_sq_raw(z11);
  fe10 z_5_0 = fe10_mul_raw(z9, z22);                           // 2^5 - 1
  fe10 z_10_0 = fe10_mul_raw(fe10_sqn(z_5_0, 5), z_5_0);        // 2^10 - 1
  fe10 z_20_0 = fe10_mul_raw(fe10_sqn(z_10_0, 10), z_10_0);     // 2^20 - 1
  fe10 z_40_0 = fe10_mul_raw(fe10_sqn(z_20_0, 20), z_20_0);     // 2^40 - 1
  fe10 z_50_0 = fe10_mul_raw(fe10_sqn(z_40_0, 10), z_10_0);     // 2^50 - 1
  fe10 z_100_0 = fe10_mul_raw(fe10_sqn(z_50_0, 50), z_50_0);    // 2^100 - 1
  fe10 z_200_0 = fe10_mul_raw(fe10_sqn(z_100_0, 100), z_100_0);
  t250 = fe10_mul_raw(fe10_sqn(z_200_0, 50), z_50_0);           // 2^250 - 1
}
