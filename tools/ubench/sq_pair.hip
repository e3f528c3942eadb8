// Does interleaving two independent squarings column by column hide the serial carry path of the sequential-carry design?
// (fe.cuh: column k's mad chain starts from column k-1's carry, so each column boundary is a dependent mad -> shift -> mad step;
// the pins that keep LLVM from re-associating the carries also keep it from interleaving two field operations by itself.)
// Result (round 2, MI355X): no.  Two raw squarings interleaved column by column take 1.51 ms against 1.50 ms one after the other
// (the second wave of the SIMD already hides the path).  The same program prices all-limb centring below even-limb centring
// (202 vs 242 units), but in the doubling chain itself all-limb centring is 1.5 % slower on C3 (same-box A/B) - chained
// microbenchmarks of single operations do not predict the in-situ schedule; only in-situ A/Bs decide.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/sq_pair.hip -o variants/sq_pair
#include "fe10_old.cuh"   /* the 10 x 25.5-bit field arithmetic of round 1, frozen for these measurements */
#include <cstdio>
#include <vector>

// two raw squarings, columns interleaved: A_k's mads, then B_k's mads, then A_k's carry step, then B_k's
template <uint32_t CMASK>
AFX_DEV void fe_sq2_impl(fe& ra, fe& rb, const fe& fa, const fe& fb) {
  int32_t a2[10], a19[10], a38[10], b2[10], b19[10], b38[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    a2[i] = (int32_t)(2u * (uint32_t)fa.v[i]); a19[i] = (int32_t)(19u * (uint32_t)fa.v[i]); a38[i] = (int32_t)(38u * (uint32_t)fa.v[i]);
    b2[i] = (int32_t)(2u * (uint32_t)fb.v[i]); b19[i] = (int32_t)(19u * (uint32_t)fb.v[i]); b38[i] = (int32_t)(38u * (uint32_t)fb.v[i]);
  }
  int64_t ca = (CMASK & 1u) ? (1LL << 25) : 0, cb = ca;
  uint32_t ua0 = 0, ub0 = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    int64_t HA = ca, HB = cb;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      if (j < i) continue;
      const bool wrap = i + j >= 10;
      const bool odd2 = (i & 1) && (j & 1);
      HA += (int64_t)((i == j) ? fa.v[i] : a2[i]) * (int64_t)(wrap ? (odd2 ? a38[j] : a19[j]) : (odd2 ? a2[j] : fa.v[j]));
      AFX_PIN(HA);
    }
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      if (j < i) continue;
      const bool wrap = i + j >= 10;
      const bool odd2 = (i & 1) && (j & 1);
      HB += (int64_t)((i == j) ? fb.v[i] : b2[i]) * (int64_t)(wrap ? (odd2 ? b38[j] : b19[j]) : (odd2 ? b2[j] : fb.v[j]));
      AFX_PIN(HB);
    }
    const int bits = (k & 1) ? 25 : 26;
    const uint32_t la = (uint32_t)HA & ((1u << bits) - 1), lb = (uint32_t)HB & ((1u << bits) - 1);
    if (k == 0) { ua0 = la; ub0 = lb; }
    else {
      ra.v[k] = ((CMASK >> k) & 1u) ? (int32_t)la - (1 << (bits - 1)) : (int32_t)la;
      rb.v[k] = ((CMASK >> k) & 1u) ? (int32_t)lb - (1 << (bits - 1)) : (int32_t)lb;
    }
    const bool cn = k < 9 && ((CMASK >> (k + 1)) & 1u);
    ca = cn ? ((HA + (1LL << 50)) >> bits) : (HA >> bits);
    cb = cn ? ((HB + (1LL << 50)) >> bits) : (HB >> bits);
  }
  int64_t A0 = (int64_t)ua0 + ca * 19, B0 = (int64_t)ub0 + cb * 19;
  ra.v[0] = (CMASK & 1u) ? (int32_t)((uint32_t)A0 & 0x3ffffffu) - (1 << 25) : (int32_t)((uint32_t)A0 & 0x3ffffffu);
  rb.v[0] = (CMASK & 1u) ? (int32_t)((uint32_t)B0 & 0x3ffffffu) - (1 << 25) : (int32_t)((uint32_t)B0 & 0x3ffffffu);
  ra.v[1] += (int32_t)(A0 >> 26);
  rb.v[1] += (int32_t)(B0 >> 26);
}

constexpr int ITERS = 2000;
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_fe(int32_t* p) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  fe a, b;
  for (int i = 0; i < 10; i++) { a.v[i] = p[i * 512 * 256 + t] & 0x1ffffff; b.v[i] = (p[i * 512 * 256 + t] >> 3) & 0xffffff; }
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (MODE == 0) { a = fe_sq_raw(a); b = fe_sq_raw(b); }
    else if constexpr (MODE == 1) { fe x, y; fe_sq2_impl<0u>(x, y, a, b); a = x; b = y; }
    else if constexpr (MODE == 2) { a = fe_sq_even(a); b = fe_sq_even(b); }
    else if constexpr (MODE == 3) { fe x, y; fe_sq2_impl<AFX_CENTRE_EVEN>(x, y, a, b); a = x; b = y; }
    else if constexpr (MODE == 4) { a = fe_mul_raw(a, b); b = fe_mul_raw(b, a); }
    else if constexpr (MODE == 5) { a = fe_sq(a); b = fe_sq(b); }
    else if constexpr (MODE == 6) { a = fe_sq_impl<0x2aau>(a); b = fe_sq_impl<0x2aau>(b); }   // odd limbs centred
    else if constexpr (MODE == 7) { a = fe_sq_raw(a); b = fe_sq_raw(b); a = fe_carry(a); }       // raw + one generic carry pass
    else if constexpr (MODE == 8) { a = fe_mul(a, b); b = fe_mul(b, a); }
  }
  for (int i = 0; i < 10; i++) p[i * 512 * 256 + t] = a.v[i] ^ b.v[i];
}
template <int MODE>
static void run(int32_t* d, const char* name, int ncu) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_fe<MODE>, dim3(ncu * 2), dim3(256), 0, 0, d);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fe<MODE>, dim3(ncu * 2), dim3(256), 0, 0, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("  %-34s %8.3f ms  per field operation: %.0f ns x 1e-3 per wave-level op per SIMD (relative figures matter)\n", name, best, best * 1e6 / (2.0 * ITERS * 2));
}
int main() {
  hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int ncu = pr.multiProcessorCount;
  int32_t* d; hipMalloc(&d, sizeof(int32_t) * 10 * 512 * 256);
  std::vector<int32_t> h(10 * 512 * 256);
  for (size_t i = 0; i < h.size(); i++) h[i] = (int32_t)(i * 2654435761u >> 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) {
    run<0>(d, "2 x fe_sq_raw, one after the other", ncu);
    run<1>(d, "fe_sq2 raw, columns interleaved", ncu);
    run<2>(d, "2 x fe_sq_even, one after the other", ncu);
    run<3>(d, "fe_sq2 even-centred, interleaved", ncu);
    run<4>(d, "2 x fe_mul_raw (dependent)", ncu);
    run<5>(d, "2 x fe_sq (all limbs centred)", ncu);
    run<6>(d, "2 x fe_sq odd limbs centred", ncu);
    run<7>(d, "2 x fe_sq_raw + 1 fe_carry", ncu);
    run<8>(d, "2 x fe_mul centred (dependent)", ncu);
  }
  return 0;
}
