// Experiment: GF(2^255-19) in 9 signed limbs of 29 bits (limb 8: 23 bits), against the shipped 10 x 25.5-bit form (fe.cuh).
// 81 multiply-adds per product instead of 100 and no premultiplied operands: the high columns 9..16 are accumulated on
// their own and folded as 1216 * low32 (column k) + 9728 * high32 (column k+1), since 2^261 = 1216 (mod p) and
// 2^32 = 8 * 2^29.  The low columns run the shipped sequential carry (carry of column k = addend of column k+1's chain).
// Units: 1 = 2^29.  Raw results: limbs in [0, 1).  Column sums: 8 full products in column 7 => |A| * |B| < 4.
#pragma once
#include <stdint.h>
#ifndef FE9_DEV
#define FE9_DEV static inline
#endif
#ifndef FE9_PIN
#define FE9_PIN(x) ((void)0)
#endif
struct fe9 { int32_t v[9]; };
#define FE9_MASK29 0x1fffffffu
#define FE9_MASK23 0x7fffffu

template <bool CENTRED>
FE9_DEV fe9 fe9_finish(int64_t (&hi)[8], const fe9& f, const fe9& g, bool square);

// low columns + folds + carry; `lowcol(k)` returns the column's own products
template <bool CENTRED, class LOW>
FE9_DEV fe9 fe9_reduce(const int64_t (&hi)[8], LOW&& lowcol) {
  fe9 r;
  int64_t c = CENTRED ? (1LL << 28) : 0;
  uint32_t u0 = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    int64_t H = c;
    H = lowcol(k, H);
    if (k < 8) { H += (int64_t)((uint64_t)(uint32_t)hi[k] * 1216u); FE9_PIN(H); }
    if (k > 0) { H += (int64_t)(int32_t)(hi[k - 1] >> 32) * 9728; FE9_PIN(H); }
    if (k < 8) {
      const uint32_t lo = (uint32_t)H & FE9_MASK29;
      if (k == 0) u0 = lo; else r.v[k] = CENTRED ? (int32_t)lo - (1 << 28) : (int32_t)lo;
      // the next limb's rounding constant travels in the carry: 2^28 for limbs 1..7, 2^22 for limb 8
      c = CENTRED ? ((H + (k < 7 ? (1LL << 57) : (1LL << 51))) >> 29) : (H >> 29);
    } else {
      const uint32_t lo = (uint32_t)H & FE9_MASK23;
      r.v[8] = CENTRED ? (int32_t)lo - (1 << 22) : (int32_t)lo;
      c = H >> 23;   // weight 2^255 = 19
    }
  }
  int64_t H0 = (int64_t)u0 + c * 19;
  const int32_t c0 = (int32_t)(H0 >> 29);
  r.v[0] = CENTRED ? (int32_t)((uint32_t)H0 & FE9_MASK29) - (1 << 28) : (int32_t)((uint32_t)H0 & FE9_MASK29);
  r.v[1] += c0;
  return r;
}

template <bool CENTRED>
FE9_DEV fe9 fe9_mul_impl(const fe9& f, const fe9& g) {
  int64_t hi[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    int64_t H = 0;
#pragma unroll
    for (int i = m + 1; i < 9; i++) { H += (int64_t)f.v[i] * (int64_t)g.v[9 + m - i]; FE9_PIN(H); }
    hi[m] = H;
  }
  return fe9_reduce<CENTRED>(hi, [&](int k, int64_t H) {
#pragma unroll
    for (int i = 0; i <= k; i++) { H += (int64_t)f.v[i] * (int64_t)g.v[k - i]; FE9_PIN(H); }
    return H;
  });
}
template <bool CENTRED>
FE9_DEV fe9 fe9_sq_impl(const fe9& f) {
  int32_t f2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) f2[i] = (int32_t)(2u * (uint32_t)f.v[i]);
  int64_t hi[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int k = 9 + m;
    int64_t H = 0;
#pragma unroll
    for (int i = m + 1; i < 9; i++) {
      const int j = k - i;
      if (j < i) continue;
      H += (int64_t)(i == j ? f.v[i] : f2[i]) * (int64_t)f.v[j];
      FE9_PIN(H);
    }
    hi[m] = H;
  }
  return fe9_reduce<CENTRED>(hi, [&](int k, int64_t H) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      const int j = k - i;
      if (j < i) continue;
      H += (int64_t)(i == j ? f.v[i] : f2[i]) * (int64_t)f.v[j];
      FE9_PIN(H);
    }
    return H;
  });
}

// variant: the high columns accumulated operand by operand (consecutive multiply-adds go to different accumulators)
template <bool CENTRED>
FE9_DEV fe9 fe9_mul_rows(const fe9& f, const fe9& g) {
  int64_t hi[8];
#pragma unroll
  for (int m = 0; m < 8; m++) hi[m] = 0;
#pragma unroll
  for (int i = 1; i < 9; i++) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int j = 9 + m - i;
      if (j < 1 || j > 8) continue;
      hi[m] += (int64_t)f.v[i] * (int64_t)g.v[j];
      FE9_PIN(hi[m]);
    }
  }
  return fe9_reduce<CENTRED>(hi, [&](int k, int64_t H) {
#pragma unroll
    for (int i = 0; i <= k; i++) { H += (int64_t)f.v[i] * (int64_t)g.v[k - i]; FE9_PIN(H); }
    return H;
  });
}
// variant: no pins at all (the compiler schedules freely)
template <bool CENTRED>
FE9_DEV fe9 fe9_mul_nopin(const fe9& f, const fe9& g) {
  int64_t hi[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    int64_t H = 0;
#pragma unroll
    for (int i = m + 1; i < 9; i++) H += (int64_t)f.v[i] * (int64_t)g.v[9 + m - i];
    hi[m] = H;
  }
  fe9 r;
  int64_t c = 0;
  uint32_t u0 = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    int64_t H = c;
#pragma unroll
    for (int i = 0; i <= k; i++) H += (int64_t)f.v[i] * (int64_t)g.v[k - i];
    if (k < 8) H += (int64_t)((uint64_t)(uint32_t)hi[k] * 1216u);
    if (k > 0) H += (int64_t)(int32_t)(hi[k - 1] >> 32) * 9728;
    if (k < 8) { const uint32_t lo = (uint32_t)H & FE9_MASK29; if (k == 0) u0 = lo; else r.v[k] = (int32_t)lo; c = H >> 29; }
    else { r.v[8] = (int32_t)((uint32_t)H & FE9_MASK23); c = H >> 23; }
  }
  int64_t H0 = (int64_t)u0 + c * 19;
  r.v[0] = (int32_t)((uint32_t)H0 & FE9_MASK29);
  r.v[1] += (int32_t)(H0 >> 29);
  return r;
}
FE9_DEV fe9 fe9_mul(const fe9& f, const fe9& g) { return fe9_mul_impl<true>(f, g); }
FE9_DEV fe9 fe9_mul_raw(const fe9& f, const fe9& g) { return fe9_mul_impl<false>(f, g); }
FE9_DEV fe9 fe9_sq(const fe9& f) { return fe9_sq_impl<true>(f); }
FE9_DEV fe9 fe9_sq_raw(const fe9& f) { return fe9_sq_impl<false>(f); }
FE9_DEV fe9 fe9_add(const fe9& a, const fe9& b) { fe9 r; for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
FE9_DEV fe9 fe9_sub(const fe9& a, const fe9& b) { fe9 r; for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
// a - p, limb-wise: limbs of a raw sum in [0, 2) come out in (-1, 1)
FE9_DEV fe9 fe9_sub_p(const fe9& a) {
  fe9 r;
  r.v[0] = a.v[0] - (int32_t)(FE9_MASK29 - 18);
  for (int i = 1; i < 8; i++) r.v[i] = a.v[i] - (int32_t)FE9_MASK29;
  r.v[8] = a.v[8] - (int32_t)FE9_MASK23;
  return r;
}
