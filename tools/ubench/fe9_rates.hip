// 9 x 29-bit limbs (fe9.h) against the shipped 10 x 25.5-bit field arithmetic (fe.cuh), at k_msm's occupancy (256 threads x 2
// blocks per CU), same binary, same run.  Build: hipcc -O3 --offload-arch=gfx950 fe9_rates.hip -o build/fe9_rates
#include "fe10_old.cuh"   /* the 10 x 25.5-bit field arithmetic of round 1, frozen for these measurements */
#define FE9_DEV __device__ __forceinline__
#define FE9_PIN(x) asm volatile("" ::"v"(x))
#include "fe9.h"
#include <cstdio>
#include <vector>

constexpr int ITERS = 2000;

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_fe(int32_t* p) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if constexpr (MODE < 10) {
    fe a, b;
    for (int i = 0; i < 10; i++) { a.v[i] = p[i * 512 * 256 + t] & 0x1ffffff; b.v[i] = (p[i * 512 * 256 + t] >> 3) & 0xffffff; }
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
      if constexpr (MODE == 0) { a = fe_mul_raw(a, b); }
      else if constexpr (MODE == 1) { a = fe_sq_raw(a); }
      else if constexpr (MODE == 2) {   // the shipped doubling (ge_p2_dbl + conversion to p2)
        fe XX = fe_sq_even(a), YY = fe_sq_even(b), ZZ = fe_sq_raw(fe_sub(a, XX));
        fe B = fe_add(ZZ, ZZ), A = fe_sub(b, a), AA = fe_sq_even(A);
        fe Y3 = fe_add(YY, XX), Z3 = fe_sub(YY, XX), X3 = fe_sub(Y3, AA), T3 = fe_sub(B, Z3);
        a = fe_mul_raw(T3, X3); b = fe_mul_raw(Y3, Z3); fe z = fe_mul_raw(T3, Z3);
        b = fe_sub(b, z);
      } else {   // the shape of an addition: 8 products
        fe A = fe_mul_raw(fe_add(b, a), b), B = fe_mul_raw(fe_sub(b, a), a), Cn = fe_mul_raw(a, b), D = fe_mul_raw(b, b);
        fe X3 = fe_sub(A, B), Y3 = fe_add(A, B), Z3 = fe_sub(D, Cn), T3 = fe_add(D, Cn);
        a = fe_mul_raw(T3, X3); b = fe_mul_raw(Y3, Z3); fe z = fe_mul_raw(T3, Z3), tt = fe_mul_raw(Y3, X3);
        a = fe_sub(a, z); b = fe_sub(b, tt);
      }
    }
    for (int i = 0; i < 10; i++) p[i * 512 * 256 + t] = a.v[i] ^ b.v[i];
  } else {
    fe9 a, b;
    for (int i = 0; i < 9; i++) { a.v[i] = p[i * 512 * 256 + t] & 0xfffffff; b.v[i] = (p[i * 512 * 256 + t] >> 3) & 0x7ffffff; }
    a.v[8] &= 0x3fffff; b.v[8] &= 0x3fffff;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
      if constexpr (MODE == 10) { a = fe9_mul_raw(a, b); }
      else if constexpr (MODE == 11) { a = fe9_sq_raw(a); }
      else if constexpr (MODE == 14) { a = fe9_mul_rows<false>(a, b); }
      else if constexpr (MODE == 15) { a = fe9_mul_nopin<false>(a, b); }
      else if constexpr (MODE == 16) {   // addition shape with the row-wise high columns
        fe9 A = fe9_mul_rows<false>(fe9_add(b, a), b), B = fe9_mul_rows<false>(fe9_sub(b, a), a), Cn = fe9_mul_rows<false>(a, b), D = fe9_mul_rows<false>(b, b);
        fe9 X3 = fe9_sub(A, B), Y3 = fe9_add(A, B), Z3 = fe9_sub(D, Cn), T3 = fe9_add(D, Cn);
        a = fe9_mul_rows<false>(T3, X3); b = fe9_mul_rows<false>(Y3, Z3); fe9 z = fe9_mul_rows<false>(T3, Z3), tt = fe9_mul_rows<false>(Y3, X3);
        a = fe9_sub(a, z); b = fe9_sub(b, tt);
      } else if constexpr (MODE == 17) {
        fe9 A = fe9_mul_nopin<false>(fe9_add(b, a), b), B = fe9_mul_nopin<false>(fe9_sub(b, a), a), Cn = fe9_mul_nopin<false>(a, b), D = fe9_mul_nopin<false>(b, b);
        fe9 X3 = fe9_sub(A, B), Y3 = fe9_add(A, B), Z3 = fe9_sub(D, Cn), T3 = fe9_add(D, Cn);
        a = fe9_mul_nopin<false>(T3, X3); b = fe9_mul_nopin<false>(Y3, Z3); fe9 z = fe9_mul_nopin<false>(T3, Z3), tt = fe9_mul_nopin<false>(Y3, X3);
        a = fe9_sub(a, z); b = fe9_sub(b, tt);
      }
      else if constexpr (MODE == 12) {   // doubling: three raw squarings, one centred, two offsets by p
        fe9 XX = fe9_sq_raw(a), YY = fe9_sq_raw(b), ZZ = fe9_sq_raw(fe9_sub(a, XX));
        fe9 B = fe9_sub_p(fe9_add(ZZ, ZZ)), A = fe9_sub(b, a), AA = fe9_sq(A);
        fe9 Y3 = fe9_sub_p(fe9_add(YY, XX)), Z3 = fe9_sub(YY, XX), X3 = fe9_sub(Y3, AA), T3 = fe9_sub(B, Z3);
        a = fe9_mul_raw(T3, X3); b = fe9_mul_raw(Y3, Z3); fe9 z = fe9_mul_raw(T3, Z3);
        b = fe9_sub(b, z);
      } else {
        fe9 A = fe9_mul_raw(fe9_add(b, a), b), B = fe9_mul_raw(fe9_sub(b, a), a), Cn = fe9_mul_raw(a, b), D = fe9_mul_raw(b, b);
        fe9 X3 = fe9_sub(A, B), Y3 = fe9_add(A, B), Z3 = fe9_sub(D, Cn), T3 = fe9_add(D, Cn);
        a = fe9_mul_raw(T3, X3); b = fe9_mul_raw(Y3, Z3); fe9 z = fe9_mul_raw(T3, Z3), tt = fe9_mul_raw(Y3, X3);
        a = fe9_sub(a, z); b = fe9_sub(b, tt);
      }
    }
    for (int i = 0; i < 9; i++) p[i * 512 * 256 + t] = a.v[i] ^ b.v[i];
  }
}

template <int MODE>
static float run(int32_t* d, const char* name, double fe_ops_per_iter, int ncu) {
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
  const double cyc = best * 1e-3 * 2.4e9 / (2.0 * ITERS * fe_ops_per_iter);
  printf("  %-22s %8.3f ms  %.0f cycles per wave-level field op per SIMD (at 2.4 GHz)\n", name, best, cyc);
  return best;
}

int main() {
  hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int ncu = pr.multiProcessorCount;
  int32_t* d; hipMalloc(&d, sizeof(int32_t) * 10 * 512 * 256);
  std::vector<int32_t> h(10 * 512 * 256);
  for (size_t i = 0; i < h.size(); i++) h[i] = (int32_t)(i * 2654435761u >> 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("10 x 25.5 bits (shipped) vs 9 x 29 bits, %d CUs, best of 5\n", ncu);
  for (int rep = 0; rep < 2; rep++) {
    const float m10 = run<0>(d, "10: mul raw", 1, ncu), m9 = run<10>(d, " 9: mul raw", 1, ncu);
    const float s10 = run<1>(d, "10: sq raw", 1, ncu), s9 = run<11>(d, " 9: sq raw", 1, ncu);
    const float d10 = run<2>(d, "10: doubling (4S+3M)", 7, ncu), d9 = run<12>(d, " 9: doubling (4S+3M)", 7, ncu);
    const float a10 = run<3>(d, "10: addition (8M)", 8, ncu), a9 = run<13>(d, " 9: addition (8M)", 8, ncu);
    run<16>(d, " 9: addition, row-wise high columns", 8, ncu);
    run<17>(d, " 9: addition, no pins", 8, ncu);
    printf("  ratios 9/10: mul %.3f  sq %.3f  doubling %.3f  addition %.3f\n", m9 / m10, s9 / s10, d9 / d10, a9 / a10);
  }
  return 0;
}
