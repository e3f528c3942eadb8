// FROZEN COPY (round 1 / early round 2) of aeonflux_amd/csrc/fe.cuh: the 10 x 25.5-bit field arithmetic, kept for the
// microbenchmarks that measured it (fe_rates, carry_variants, sq_pair, fe9_rates).  Not part of the product.
// GF(2^255-19) for gfx950, one field element per lane.
//
// Representation: 10 signed limbs, radix 2^25.5 (26,25,26,25,... bits).  Chosen by measurement
// (profiles/r01_valu_rates_ubench.txt, at the 2 waves per SIMD the kernels run at): v_mad_i64_i32 issues at ~5.5 cycles
// per wave-instruction, v_fma_f64 at ~5.1, plain 32-bit VOP2 ops at ~2.8, VOP3 forms and 64-bit shifts at ~4.7, so the
// 100-mad schoolbook with 64-bit column accumulators and NO carry handling inside the accumulation beats every
// fp64-split and saturated-limb variant priced against it.
// One fe = 10 VGPRs.  Replaces, for the reference's call sites, what curve25519-dalek's
// FieldElement [3P] does (e.g. under /root/reference/src/nizk/presentation.rs:342-351); only canonical
// encodings are contractual.
//
// The kernels built on this file are VALU-issue bound and their time is the sum of per-opcode issue costs
// (profiles/r01_valu_rates_ubench.txt, profiles/r01_fe_rates_ubench.txt), so the code below is written against
// that price list: 64-bit adds (v_lshl_add_u64, ~6.3 cycles) are avoided by feeding each column's carry into
// the next column's mad chain as its addend.  Measured: fe_mul 753 cycles per wave-level operation (617 raw),
// fe_sq 552 (488 raw; 56 mads + 40 other VALU instructions).
//
// Bounds discipline (same as the classic 10-limb schedule): fe_sq and fe_mul's SECOND operand accept
// limbs up to 1.65*2^26 (even) / 1.65*2^25 (odd) in magnitude (the 19x / 38x premultiplications must
// fit int32); fe_mul's FIRST operand may be up to 4*2^26 / 4*2^25 (column sums stay < 2^63: worst
// column = 124.5 * F * G * 2^52).  Results are within 1.01*2^25 / 1.01*2^24.
// fe_add/fe_sub/fe_neg are limb-wise with no carry; at most one add/sub level of reduced operands
// (or the documented three-term sums) may feed a multiplication.  fe_carry() re-normalises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AFX_DEV __device__ __forceinline__

struct fe {
  int32_t v[10];
};

AFX_DEV fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = 0;
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
  for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
AFX_DEV fe fe_sub(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = a.v[i] - b.v[i];
  return r;
}
AFX_DEV fe fe_neg(const fe& a) {
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = -a.v[i];
  return r;
}
// f = b ? g : f
AFX_DEV void fe_cmov(fe& f, const fe& g, bool b) {
#pragma unroll
  for (int i = 0; i < 10; i++) f.v[i] = b ? g.v[i] : f.v[i];
}
AFX_DEV void fe_cswap(fe& f, fe& g, bool b) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    int32_t x = f.v[i], y = g.v[i];
    f.v[i] = b ? y : x;
    g.v[i] = b ? x : y;
  }
}

// carry chain over 64-bit column sums -> reduced 32-bit limbs
AFX_DEV fe fe_carry64(int64_t h[10]) {
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
  fe r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = (int32_t)h[i];
  return r;
}

// re-normalise a lazily added value (any limbs that fit int32)
AFX_DEV fe fe_carry(const fe& f) {
  int64_t h[10];
#pragma unroll
  for (int i = 0; i < 10; i++) h[i] = f.v[i];
  return fe_carry64(h);
}

// Pins a partial sum: the volatile (input-only, empty) statement forces the value to exist at this point, which keeps
// LLVM's reassociation from pulling the carry out of the mad chain into a separate 64-bit add.  It emits no code and,
// having no outputs, triggers none of the hazard no-ops the compiler puts after inline-asm definitions.
#if defined(__HIPCC__)
#define AFX_PIN(x) asm volatile("" ::"v"(x))
#else
#define AFX_PIN(x) ((void)0)   // host build of this header (tests/hostsim/arith_host.cpp)
#endif
// operation counters for the host build (the per-item counts DESIGN.md publishes are measured with them)
// AFX_CHECK_BOUNDS (host build only): every multiplication / squaring checks what its code relies on - the int32
// premultiplications and the 64-bit column sums - on the actual operands, and reports a violation.
#ifdef AFX_CHECK_BOUNDS
extern "C" void afx_bounds_violation(const char* what);
static inline void afx_check_products(const int32_t* f, const int32_t* g, bool square) {
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

// Schoolbook product, columns in order 0..9: column k's mad chain starts from the carry out of column k-1 (the
// mad's 64-bit addend), so the carry chain needs no 64-bit additions.  CENTRED: each carry arrives with the next
// limb's rounding constant already in it (2^50 added to the high dword before the shift), which makes every limb
// come out centred: r_k = (H_k mod 2^b) - 2^(b-1), |r_k| <= 2^(b-1).  Not CENTRED ("raw"): floor carries, limbs in
// [0, 2^b): 18 fewer additions, for results whose consumer is known to tolerate twice the magnitude (below).
// CMASK: bit k set = limb k comes out centred (its rounding constant travels in the carry of column k-1), clear = raw.
template <uint32_t CMASK>
AFX_DEV fe fe_mul_impl(const fe& f, const fe& g) {
  AFX_COUNT(afx_n_mul);
  AFX_CHECK_MUL(f, g);
  int32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    g19[i] = (int32_t)(19u * (uint32_t)g.v[i]);
    f2[i] = (int32_t)(2u * (uint32_t)f.v[i]);
  }
  fe r;
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
      AFX_PIN(H);
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
#define AFX_CENTRE_ALL 0x3ffu
#define AFX_CENTRE_EVEN 0x154u   /* limbs 2, 4, 6, 8 */
AFX_DEV fe fe_mul(const fe& f, const fe& g) { return fe_mul_impl<AFX_CENTRE_ALL>(f, g); }
// Raw result: limbs in [0, 2^26) / [0, 2^25) ("1 unit" where a centred result is 1/2 unit).  Valid as either operand
// of a multiplication or as the input of a squaring; sums of two raw values (2 units) only as a FIRST operand; a
// difference of two raw values (+-1 unit) anywhere.  ge.cuh documents, at each use, why the consumer qualifies.
AFX_DEV fe fe_mul_raw(const fe& f, const fe& g) { return fe_mul_impl<0u>(f, g); }

// The same two flavours for the squaring (raw: the squaring chains of the inversions, and Z^2 of the doubling).
template <uint32_t CMASK>
AFX_DEV fe fe_sq_impl(const fe& f) {
  AFX_COUNT(afx_n_sq);
  AFX_CHECK_SQ(f);
  int32_t f2[10], f19[10], f38[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    f2[i] = (int32_t)(2u * (uint32_t)f.v[i]);
    f19[i] = (int32_t)(19u * (uint32_t)f.v[i]);
    f38[i] = (int32_t)(38u * (uint32_t)f.v[i]);
  }
  fe r;
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
      AFX_PIN(H);
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
AFX_DEV fe fe_sq(const fe& f) { return fe_sq_impl<AFX_CENTRE_ALL>(f); }
AFX_DEV fe fe_sq_raw(const fe& f) { return fe_sq_impl<0u>(f); }
// Only the limbs whose 19-fold must fit int32 when the value is a SECOND operand - the even limbs 2, 4, 6, 8 (limb 0 is
// never premultiplied, odd limbs have a bit to spare) - come out centred.  For values that are combined with one or
// two others of their kind and then used as a second operand, never squared: XX, YY and (Y-X)^2 of the doubling, which
// meet in X3 = (YY + XX) - (Y-X)^2 and Z3 = YY - XX.
AFX_DEV fe fe_sq_even(const fe& f) { return fe_sq_impl<AFX_CENTRE_EVEN>(f); }

// f^(2^n), n >= 1, rolled loop (keeps the inversion chains small in code size); every consumer multiplies the result
AFX_DEV fe fe_sqn(fe f, int n) {
#pragma unroll 1
  for (int i = 0; i < n; i++) f = fe_sq_raw(f);
  return f;
}

// multiply by a small constant limb-set given as a reduced fe in constant memory: just fe_mul.

// Load from 8 little-endian dwords, ignoring bit 255 (dalek FieldElement::from_bytes).
// Limbs come out unsigned (< 2^26 / 2^25): inside fe_mul's input bounds, no carry needed.
AFX_DEV fe fe_frombytes(const uint32_t w[8]) {
  fe r;
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

// Canonical little-endian encoding into 8 dwords.  Input: any limbs fe_carry accepts.
AFX_DEV void fe_tobytes(uint32_t w[8], const fe& f) {
  fe t = fe_carry(f);
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
  for (int i = 0; i < 10; i++) r.v[i] = b ? -f.v[i] : f.v[i];
  return r;
}
AFX_DEV fe fe_abs(const fe& f) { return fe_cneg(f, fe_is_negative(f)); }

// z^(2^250-1) and z^11
AFX_DEV void fe_pow22501(fe& t250, fe& z11, const fe& z) {
  fe z2 = fe_sq(z);
  fe z8 = fe_sqn(z2, 2);
  fe z9 = fe_mul(z, z8);
  z11 = fe_mul(z2, z9);
  fe z22 = fe_sq(z11);
  fe z_5_0 = fe_mul(z9, z22);                       // 2^5 - 1
  fe z_10_0 = fe_mul(fe_sqn(z_5_0, 5), z_5_0);      // 2^10 - 1
  fe z_20_0 = fe_mul(fe_sqn(z_10_0, 10), z_10_0);   // 2^20 - 1
  fe z_40_0 = fe_mul(fe_sqn(z_20_0, 20), z_20_0);   // 2^40 - 1
  fe z_50_0 = fe_mul(fe_sqn(z_40_0, 10), z_10_0);   // 2^50 - 1
  fe z_100_0 = fe_mul(fe_sqn(z_50_0, 50), z_50_0);  // 2^100 - 1
  fe z_200_0 = fe_mul(fe_sqn(z_100_0, 100), z_100_0);
  t250 = fe_mul(fe_sqn(z_200_0, 50), z_50_0);       // 2^250 - 1
}
AFX_DEV fe fe_invert(const fe& z) {
  fe t250, z11;
  fe_pow22501(t250, z11, z);
  return fe_mul(fe_sqn(t250, 5), z11);  // 2^255 - 21
}
AFX_DEV fe fe_pow22523(const fe& z) {
  fe t250, z11;
  fe_pow22501(t250, z11, z);
  return fe_mul(fe_sqn(t250, 2), z);    // 2^252 - 3
}
