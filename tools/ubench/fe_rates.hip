// Field-multiplication throughput on gfx950 at the occupancy k_msm runs at (256 threads x 2 blocks per CU).
// Build:  hipcc -O3 --offload-arch=gfx950 fe_rates.hip -o build/fe_rates
#include "fe10_old.cuh"   /* the 10 x 25.5-bit field arithmetic of round 1, frozen for these measurements */
#include <cstdio>
#include <vector>

constexpr int ITERS = 2000;

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_fe(int32_t* p) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  fe a, b;
  for (int i = 0; i < 10; i++) { a.v[i] = p[i * 512 * 256 + t] & 0x1ffffff; b.v[i] = (p[i * 512 * 256 + t] >> 3) & 0xffffff; }
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (MODE == 0) { a = fe_mul(a, b); }
    else if constexpr (MODE == 1) { a = fe_sq(a); }
    else if constexpr (MODE == 3) { a = fe_mul_raw(a, b); }
    else if constexpr (MODE == 4) { a = fe_sq_raw(a); }
    else {  // the shape of a doubling: 4 squarings, sums, 3 products
      fe xx = fe_sq(a), yy = fe_sq(b), s = fe_sq(fe_add(a, b)), zz = fe_sq(fe_sub(a, b));
      fe h = fe_add(yy, xx), g = fe_sub(yy, xx), e = fe_sub(s, h), f = fe_sub(fe_add(zz, zz), g);
      a = fe_mul(e, f); b = fe_mul(g, h);
    }
  }
  for (int i = 0; i < 10; i++) p[i * 512 * 256 + t] = a.v[i] ^ b.v[i];
}

template <int MODE>
static void run(int32_t* d, const char* name, double fe_ops_per_iter, int ncu) {
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
  const double ops = (double)ncu * 2 * 256 * ITERS * fe_ops_per_iter;
  // cycles per wave-level field op per SIMD at 2.4 GHz: each SIMD hosts 2 waves
  const double cyc = best * 1e-3 * 2.4e9 / (2.0 * ITERS * fe_ops_per_iter);
  printf("  %-10s %8.3f ms  %.2f G field-ops/s  %.0f cycles per wave-level op per SIMD (2.4 GHz)\n", name, best, ops / best / 1e6, cyc);
}

int main() {
  hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int ncu = pr.multiProcessorCount;
  int32_t* d; hipMalloc(&d, sizeof(int32_t) * 10 * 512 * 256);
  std::vector<int32_t> h(10 * 512 * 256);
  for (size_t i = 0; i < h.size(); i++) h[i] = (int32_t)(i * 2654435761u >> 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("fe_mul / fe_sq (centred) and their raw variants, %d CUs\n", ncu);
  run<0>(d, "mul", 1, ncu);
  run<1>(d, "sq", 1, ncu);
  run<3>(d, "mul raw", 1, ncu);
  run<4>(d, "sq raw", 1, ncu);
  run<2>(d, "dbl-mix", 6, ncu);
  return 0;
}
