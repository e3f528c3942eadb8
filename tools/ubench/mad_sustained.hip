// Sustained v_mad_i64_i32 throughput with the clock it is sustained at (round 2), operands switching on every instruction.  tools/ubench/valu_rates.hip times bursts of
// 0.15-0.6 ms; the engine's kernels run for 10-160 ms at the socket power cap, so the figure to hold them against is what a
// PURE multiply-add loop sustains for that long, and the clock it does so at (shader-clock counter / constant 100 MHz counter).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mad_sustained.hip -o variants/mad_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int NCH = 8;

template <int SIGNED>
__global__ void __launch_bounds__(256, 2) k_mad(uint64_t* out, unsigned long long* probe, uint32_t seed, int iters) {
  const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  const bool p = blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long c0 = 0, r0 = 0;
  if (p) { c0 = clock64(); r0 = wall_clock64(); }
  uint64_t a[NCH];
  int32_t xs[NCH], ys[NCH];
  uint32_t x = seed * 2654435761u + t, y = (seed ^ t) * 40503u | 1u;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    a[c] = ((uint64_t)(x + c) << 32) | (y + 3 * c);
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    xs[c] = (int32_t)(x & 0x3ffffff) - (SIGNED ? (1 << 25) : 0);    // 26-bit limbs: centred (signed) or raw (unsigned), as in fe.cuh
    y ^= y << 13; y ^= y >> 17; y ^= y << 5;
    ys[c] = (int32_t)(y & 0x3ffffff) - (SIGNED ? (1 << 25) : 0);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // consecutive multiply-adds see different operand pairs, as in a schoolbook product: the multiplier array's inputs
      // switch on every instruction (the same pair eight times in a row draws far less power and stays at the nominal clock)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        a[c] = (uint64_t)((int64_t)xs[(c + k) % NCH] * (int64_t)ys[(c + 3 * k + 1) % NCH] + (int64_t)a[c]);
        asm volatile("" ::"v"(a[c]));
      }
    }
    xs[it & 7 ? 0 : 1] ^= (int32_t)(it & 0xffff);   // one cheap instruction per 64 multiply-adds keeps the values moving
  }
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s ^= a[c];
  out[t] = s;
  if (p) { probe[0] = clock64() - c0; probe[1] = wall_clock64() - r0; }
}

int main() {
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int ncu = pr.multiProcessorCount, blocks = ncu * 2;   // 2 blocks of 256 per CU = 2 waves per SIMD, as the engine runs
  uint64_t* out; unsigned long long* probe;
  hipMalloc(&out, sizeof(uint64_t) * blocks * 256); hipMalloc(&probe, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("# v_mad_i64_i32, %d CUs, 2 waves per SIMD, 8 independent chains per lane, operands changing\n", ncu);
  for (int mode = 0; mode < 4; mode++)
  for (int iters : { 2000, 20000, 200000, 600000 }) {
    if (mode >= 1 && iters < 600000) continue;
    const int sgn = mode == 0 || mode == 2;
    float best = 1e30f; unsigned long long pv[2] = { 0, 0 };
    for (int r = 0; r < 2; r++) {
      hipEventRecord(e0);
      if (sgn) hipLaunchKernelGGL(k_mad<1>, dim3(blocks), dim3(256), 0, 0, out, probe, 7u + r, iters);
      else hipLaunchKernelGGL(k_mad<0>, dim3(blocks), dim3(256), 0, 0, out, probe, 7u + r, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) { best = ms; hipMemcpy(pv, probe, 16, hipMemcpyDeviceToHost); }
    }
    const double mads = (double)blocks * 256 * iters * 8 * NCH;
    const double mhz = pv[1] ? 100.0 * pv[0] / pv[1] : 0;
    // per SIMD: 2 waves x iters*64 wave-instructions of mad (the ~2 scalar-ish VALU ops per 8 mads are counted out below)
    const double cyc = best * 1e-3 * mhz * 1e6 / (2.0 * iters * 8 * NCH);
    printf("%s limbs  kernel %9.3f ms   %6.2f T mads/s   core clock %7.1f MHz   %.2f cycles per wave-instruction per SIMD at that clock\n",
           sgn ? "centred (signed) " : "raw (unsigned)   ", best, mads / best / 1e9, mhz, cyc);
  }
  return 0;
}
