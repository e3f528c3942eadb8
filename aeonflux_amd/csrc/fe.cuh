// GF(2^255-19) for gfx950, one field element per lane.
//
// Representation: 9 signed limbs of 29 bits (limb 8: 23 bits), value = sum v[i] * 2^(29 i).  One fe = 9 VGPRs.
// Replaces, for the reference's call sites, what curve25519-dalek's FieldElement [3P] does (e.g. under
// /root/reference/src/nizk/presentation.rs:342-351); only canonical encodings are contractual.
//
// Why this form (tools/ubench/fe9_rates.hip, profiles/r02_fe9_rates.txt, against the 10 x 25.5-bit form of round 1,
// tools/ubench/fe10_old.cuh): the kernels built on this file are bound by VALU issue, every VALU instruction costs about what a
// v_mad_i64_i32 costs (profiles/r01_valu_rates_ubench.txt), so what counts is the number of instructions per product.  With a
// uniform radix there are no doubled or 19-fold copies of the operands to prepare (14 instructions per product in the 25.5-bit
// form): the product is 81 multiply-adds; its high columns 9..16 are accumulated on their own and folded into the low ones with
// two multiply-adds each - 2^261 = 1216 (mod p) and 2^32 = 8 * 2^29, so column k+9 = hi * 2^32 + lo adds 1216 * lo to
// column k and 9728 * hi to column k+1 - and the low columns run a sequential carry (the carry out of column k is the 64-bit
// addend of column k+1's first multiply-add, so the chain needs no 64-bit additions).  Limb 8 is cut at 23 bits and what lies
// above (weight 2^255 = 19) goes back to limb 0 with the last carry.  Measured: an addition's 8 products -9 %, a doubling's
// 4 squarings + 3 products -4 %; a lone squaring +3 % (62 multiply-adds against 56).
//
// Bounds discipline, in units of 2^29 ("1 unit"; limb 8: 2^23).  A RAW result has limbs in [0, 1); a CENTRED one in
// [-1/2, 1/2] (its rounding constants travel in the carries: 18 more additions, on the serial path).  fe_add/fe_sub/fe_neg
// are limb-wise with no carry, so magnitudes add.  A product needs |f| * |g| <= 3.8 (column 7 holds 8 full products:
// 8 * 3.8 * 2^58 plus the folds stays below 2^63), a squaring therefore |f| <= 1.9 (its doubled copy then fits int32).  ge.cuh states,
// at each use, why the operands qualify; tests/test_device_arith_on_host.py runs the engine's chains on the host build of
// this header with every column sum checked (AFX_CHECK_BOUNDS).  fe_carry() re-normalises to centred limbs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AFX_DEV __device__ __forceinline__
#define AFX_FE_LIMBS 9
#define AFX_FE_MASK29 0x1fffffffu
#define AFX_FE_MASK23 0x7fffffu

struct fe {
  int32_t v[AFX_FE_LIMBS];
};

AFX_DEV fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) r.v[i] = 0;
  return r;
}
AFX_DEV fe fe_one() {
  fe r = fe_zero();
  r.v[0] = 1;
  return r;
}
AFX_DEV fe fe_add(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
AFX_DEV fe fe_sub(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) r.v[i] = a.v[i] - b.v[i];
  return r;
}
AFX_DEV fe fe_neg(const fe& a) {
  fe r;
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) r.v[i] = -a.v[i];
  return r;
}
// a - p, limb by limb (p = 2^255 - 19 has the limbs 2^29 - 19, 2^29 - 1 (x7), 2^23 - 1): the same field element, with the
// limbs of a sum of two raw values, [0, 2), brought to (-1, 1)
AFX_DEV fe fe_sub_p(const fe& a) {
  fe r;
  r.v[0] = a.v[0] - (int32_t)(AFX_FE_MASK29 - 18);
#pragma unroll
  for (int i = 1; i < 8; i++) r.v[i] = a.v[i] - (int32_t)AFX_FE_MASK29;
  r.v[8] = a.v[8] - (int32_t)AFX_FE_MASK23;
  return r;
}
// f = b ? g : f
AFX_DEV void fe_cmov(fe& f, const fe& g, bool b) {
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) f.v[i] = b ? g.v[i] : f.v[i];
}
AFX_DEV void fe_cswap(fe& f, fe& g, bool b) {
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) {
    int32_t x = f.v[i], y = g.v[i];
    f.v[i] = b ? y : x;
    g.v[i] = b ? x : y;
  }
}

// carry chain over 64-bit limb values -> centred 32-bit limbs (|v[i]| <= 2^28, |v[8]| <= 2^22, v[1] a carry more)
AFX_DEV fe fe_carry64(int64_t h[AFX_FE_LIMBS]) {
  int64_t c;
#pragma unroll
  for (int i = 0; i < 8; i++) { c = (h[i] + (1LL << 28)) >> 29; h[i + 1] += c; h[i] -= c << 29; }
  c = (h[8] + (1LL << 22)) >> 23; h[0] += c * 19; h[8] -= c << 23;
  c = (h[0] + (1LL << 28)) >> 29; h[1] += c; h[0] -= c << 29;
  fe r;
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) r.v[i] = (int32_t)h[i];
  return r;
}

// re-normalise a lazily added value (any limbs that fit int32)
AFX_DEV fe fe_carry(const fe& f) {
  int64_t h[AFX_FE_LIMBS];
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) h[i] = f.v[i];
  return fe_carry64(h);
}

// Pins a partial sum: the volatile (input-only, empty) statement forces the value to exist at this point, which keeps
// LLVM's reassociation from pulling the carry out of the mad chain into a separate 64-bit add.  It emits no code and,
// having no outputs, triggers none of the hazard no-ops the compiler puts after inline-asm definitions.
#if defined(__HIPCC__)
#define AFX_PIN(x) asm volatile("" ::"v"(x))
#define AFX_PIN_UNIFORM(x) asm volatile("" ::"s"(x))   // the same for a wave-uniform value (a scalar register)
#else
#define AFX_PIN(x) ((void)0)   // host build of this header (tests/hostsim/arith_host.cpp)
#define AFX_PIN_UNIFORM(x) ((void)0)
#endif
// operation counters for the host build (the per-item counts DESIGN.md publishes are measured with them)
// AFX_CHECK_BOUNDS (host build only): every multiplication / squaring checks what its code relies on - the int32
// premultiplications and the 64-bit column sums - on the actual operands, and reports a violation.
#ifdef AFX_CHECK_BOUNDS
extern "C" void afx_bounds_violation(const char* what);
static inline void afx_check_products(const int32_t* f, const int32_t* g, bool square) {
  typedef __int128 i128;
  if (square)
    for (int i = 0; i < 9; i++) {
      const int64_t af = f[i] < 0 ? -(int64_t)f[i] : f[i];
      if (af * 2 >= (1LL << 31)) afx_bounds_violation("doubled limb of a squaring overflows int32");
    }
  // magnitudes, worst signs: high columns, then each low column with its own products, both folds and the carry
  i128 hi[8];
  for (int m = 0; m < 8; m++) {
    i128 sum = 0;
    for (int i = m + 1; i < 9; i++) {
      const i128 af = f[i] < 0 ? -(i128)f[i] : f[i], ag = g[9 + m - i] < 0 ? -(i128)g[9 + m - i] : g[9 + m - i];
      sum += af * ag;
    }
    if (sum >= ((i128)1 << 63) - ((i128)1 << 58)) afx_bounds_violation("high column sum beyond 2^63 - 2^58");
    hi[m] = sum;
  }
  i128 carry = 0;
  for (int k = 0; k < 9; k++) {
    i128 sum = carry + ((i128)1 << 57);
    for (int i = 0; i <= k; i++) {
      const i128 af = f[i] < 0 ? -(i128)f[i] : f[i], ag = g[k - i] < 0 ? -(i128)g[k - i] : g[k - i];
      sum += af * ag;
    }
    if (k < 8) sum += (i128)0xffffffffu * 1216;
    if (k > 0) sum += ((hi[k - 1] >> 32) + 1) * 9728;
    if (sum >= ((i128)1 << 63) - ((i128)1 << 58)) afx_bounds_violation("low column sum beyond 2^63 - 2^58");
    carry = (sum >> (k < 8 ? 29 : 23)) + 1;
  }
}
#define AFX_CHECK_MUL(f, g) afx_check_products((f).v, (g).v, false)
#define AFX_CHECK_SQ(f) afx_check_products((f).v, (f).v, true)
#else
#define AFX_CHECK_MUL(f, g) ((void)0)
#define AFX_CHECK_SQ(f) ((void)0)
#endif
#ifdef AFX_COUNT_OPS
extern thread_local uint64_t afx_n_mul, afx_n_sq;
#define AFX_COUNT(x) (++(x))
#else
#define AFX_COUNT(x) ((void)0)
#endif

// Low columns 0..8 of a product or a squaring, given its high columns: lowcol(k, H) adds column k's own products to H.
// CENTRED: each carry arrives with the next limb's rounding constant already in it (2^57 added before the shift by 29;
// 2^51 for limb 8), which makes every limb come out centred: r_k = (H_k mod 2^b) - 2^(b-1).  Not CENTRED ("raw"): floor
// carries, limbs in [0, 2^b): 17 fewer additions, and none on the serial path (last multiply-add of column k -> shift ->
// first multiply-add of column k+1).
template <bool CENTRED, class LOW>
AFX_DEV fe fe_reduce_columns(const int64_t (&hi)[8], LOW&& lowcol) {
  fe r;
  int64_t c = CENTRED ? (1LL << 28) : 0;  // rounding constant of limb 0; later carries arrive with the next limb's folded in
  uint32_t u0 = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    int64_t H = lowcol(k, c);
    if (k < 8) { H += (int64_t)((uint64_t)(uint32_t)hi[k] * 1216u); AFX_PIN(H); }        // column k+9, low dword
    if (k > 0) { H += (int64_t)(int32_t)(hi[k - 1] >> 32) * 9728; AFX_PIN(H); }          // column k+8, high dword
    if (k < 8) {
      const uint32_t lo = (uint32_t)H & AFX_FE_MASK29;
      if (k == 0) u0 = lo; else r.v[k] = CENTRED ? (int32_t)lo - (1 << 28) : (int32_t)lo;
      c = CENTRED ? ((H + (k < 7 ? (1LL << 57) : (1LL << 51))) >> 29) : (H >> 29);
    } else {
      const uint32_t lo = (uint32_t)H & AFX_FE_MASK23;
      r.v[8] = CENTRED ? (int32_t)lo - (1 << 22) : (int32_t)lo;
      c = H >> 23;   // weight 2^255 = 19
    }
  }
  // wrap: limb 0 gets 19 * (what lies above bit 255); u0 still holds limb 0 (with its rounding constant when centred)
  int64_t H0 = (int64_t)u0 + c * 19;
  const int32_t c0 = (int32_t)(H0 >> 29);
  r.v[0] = CENTRED ? (int32_t)((uint32_t)H0 & AFX_FE_MASK29) - (1 << 28) : (int32_t)((uint32_t)H0 & AFX_FE_MASK29);
  r.v[1] += c0;
  return r;
}
template <bool CENTRED>
AFX_DEV fe fe_mul_impl(const fe& f, const fe& g) {
  AFX_COUNT(afx_n_mul);
  AFX_CHECK_MUL(f, g);
  int64_t hi[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    int64_t H = 0;
#pragma unroll
    for (int i = m + 1; i < 9; i++) {
      H += (int64_t)f.v[i] * (int64_t)g.v[9 + m - i];
      // (not the chain's first product: pinned, it is computed once for the pin and once more inside the reassociated sum - a dead
      // multiply-add per high column, 243 of the windowed kernel's 5181 until round 6)
      if (i > m + 1) AFX_PIN(H);
    }
    hi[m] = H;
  }
  return fe_reduce_columns<CENTRED>(hi, [&](int k, int64_t H) {
#pragma unroll
    for (int i = 0; i <= k; i++) { H += (int64_t)f.v[i] * (int64_t)g.v[k - i]; if (k > 0) AFX_PIN(H); }   // (column 0 starts from a constant: see above)
    return H;
  });
}
AFX_DEV fe fe_mul(const fe& f, const fe& g) { return fe_mul_impl<true>(f, g); }
// Raw result: limbs in [0, 1) instead of [-1/2, 1/2].  ge.cuh documents, at each use, why the consumer tolerates it.
AFX_DEV fe fe_mul_raw(const fe& f, const fe& g) { return fe_mul_impl<false>(f, g); }

// The same two flavours for the squaring: 45 products (cross terms from a doubled copy of the operand) + the 16 folds.
template <bool CENTRED>
AFX_DEV fe fe_sq_impl(const fe& f) {
  AFX_COUNT(afx_n_sq);
  AFX_CHECK_SQ(f);
  int32_t f2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) f2[i] = (int32_t)(2u * (uint32_t)f.v[i]);
  int64_t hi[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    int64_t H = 0;
    bool first = true;
#pragma unroll
    for (int i = m + 1; i < 9; i++) {
      const int j = 9 + m - i;
      if (j < i) continue;
      H += (int64_t)(i == j ? f.v[i] : f2[i]) * (int64_t)f.v[j];
      if (!first) AFX_PIN(H);   // (as in fe_mul_impl)
      first = false;
    }
    hi[m] = H;
  }
  return fe_reduce_columns<CENTRED>(hi, [&](int k, int64_t H) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      const int j = k - i;
      if (j < i) continue;
      H += (int64_t)(i == j ? f.v[i] : f2[i]) * (int64_t)f.v[j];
      if (k > 0) AFX_PIN(H);
    }
    return H;
  });
}
AFX_DEV fe fe_sq(const fe& f) { return fe_sq_impl<true>(f); }
AFX_DEV fe fe_sq_raw(const fe& f) { return fe_sq_impl<false>(f); }

// Load from 8 little-endian dwords, ignoring bit 255 (dalek FieldElement::from_bytes).  Limbs come out raw.
AFX_DEV fe fe_frombytes(const uint32_t w[8]) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int o = 29 * i, k = o >> 5, sh = o & 31;
    uint64_t x = w[k];
    if (sh + 29 > 32 && k + 1 < 8) x |= (uint64_t)w[k + 1] << 32;
    r.v[i] = (int32_t)((uint32_t)(x >> sh) & (i < 8 ? AFX_FE_MASK29 : AFX_FE_MASK23));   // limb 8: bits 232..254
  }
  return r;
}

// Canonical little-endian encoding into 8 dwords.  Input: any limbs fe_carry accepts.
AFX_DEV void fe_tobytes(uint32_t w[8], const fe& f) {
  fe t = fe_carry(f);
  int32_t h[9];
#pragma unroll
  for (int i = 0; i < 9; i++) h[i] = t.v[i];
  // q = floor(value / p), -1 or 0 for the centred limbs fe_carry leaves (|value| < 2^254 * 1.01), by the classic estimate:
  // floor(h / p) = floor((h + 19 h / 2^255 + 1/2) / 2^255), with 19 h / 2^255 taken from the top limb and rounded
  int32_t q = (19 * h[8] + (1 << 22)) >> 23;
#pragma unroll
  for (int i = 0; i < 8; i++) q = (h[i] + q) >> 29;
  q = (h[8] + q) >> 23;
  h[0] += 19 * q;
  int32_t c;
#pragma unroll
  for (int i = 0; i < 8; i++) { c = h[i] >> 29; h[i + 1] += c; h[i] -= c << 29; }
  c = h[8] >> 23; h[8] -= c << 23;   // the dropped carry is q * 2^255: the output is value - q * p, in [0, p)
#pragma unroll
  for (int k = 0; k < 8; k++) {
    // dword k = bits 32k .. 32k+31: from limb i0 = floor(32k / 29) onwards
    const int i0 = (32 * k) / 29, sh = 32 * k - 29 * i0;
    uint64_t x = (uint64_t)(uint32_t)h[i0] >> sh;
    if (i0 + 1 < 9) x |= (uint64_t)(uint32_t)h[i0 + 1] << (29 - sh);
    if (58 - sh < 32 && i0 + 2 < 9) x |= (uint64_t)(uint32_t)h[i0 + 2] << (58 - sh);
    w[k] = (uint32_t)x;
  }
}

AFX_DEV bool fe_is_negative(const fe& f) {
  uint32_t w[8];
  fe_tobytes(w, f);
  return (w[0] & 1) != 0;
}
AFX_DEV bool fe_is_zero(const fe& f) {
  uint32_t w[8];
  fe_tobytes(w, f);
  return (w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0;
}
AFX_DEV bool fe_eq(const fe& f, const fe& g) { return fe_is_zero(fe_sub(f, g)); }
AFX_DEV fe fe_cneg(const fe& f, bool b) {
  fe r;
#pragma unroll
  for (int i = 0; i < AFX_FE_LIMBS; i++) r.v[i] = b ? -f.v[i] : f.v[i];
  return r;
}
AFX_DEV fe fe_abs(const fe& f) { return fe_cneg(f, fe_is_negative(f)); }

// The inversion and square-root chains run in the 10 x 25.5-bit form (fe10.cuh: its squaring is 5 % cheaper and a chain is
// 254 / 251 squarings and 11 products), entered and left through the canonical encoding; the result comes back centred.
#include "fe10.cuh"
AFX_DEV fe10 fe10_from_fe(const fe& z) {
  uint32_t w[8];
  fe_tobytes(w, z);
  return fe10_frombytes(w);
}
AFX_DEV fe fe_from_fe10(const fe10& z) {
  uint32_t w[8];
  fe10_tobytes(w, z);
  return fe_carry(fe_frombytes(w));
}
AFX_DEV fe fe_invert(const fe& z) {
  fe10 t250, z11;
  fe10_pow22501(t250, z11, fe10_from_fe(z));
  return fe_from_fe10(fe10_mul_raw(fe10_sqn(t250, 5), z11));  // 2^255 - 21
}
AFX_DEV fe fe_pow22523(const fe& z) {
  const fe10 z10 = fe10_from_fe(z);
  fe10 t250, z11;
  fe10_pow22501(t250, z11, z10);
  return fe_from_fe10(fe10_mul_raw(fe10_sqn(t250, 2), z10));    // 2^252 - 3
}
