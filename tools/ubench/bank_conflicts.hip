// Does v_mad_i64_i32 on gfx950 pay for VGPR bank conflicts among its source operands?  (bank = register index mod 4)
// Hard-coded registers: accumulators v[8:9], v[12:13], v[16:17], v[20:21] (banks 0,1), multiplicands chosen per case.
// Build: hipcc -O3 --offload-arch=gfx950 bank_conflicts.hip -o build/bank_conflicts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITERS = 4096;
#define MAD4(A, B) \
  "v_mad_i64_i32 v[8:9], vcc, " A ", " B ", v[8:9]\n\t"   \
  "v_mad_i64_i32 v[12:13], vcc, " A ", " B ", v[12:13]\n\t" \
  "v_mad_i64_i32 v[16:17], vcc, " A ", " B ", v[16:17]\n\t" \
  "v_mad_i64_i32 v[20:21], vcc, " A ", " B ", v[20:21]\n\t"
template <int CASE>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t x = seed * 2654435761u + t;
  asm volatile("v_mov_b32 v2, %0\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\tv_mov_b32 v6, %0\n\tv_mov_b32 v7, %0\n\t"
               "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0"
               :: "v"(x) : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21");
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (CASE == 0) asm volatile(MAD4("v2", "v3") MAD4("v6", "v7") ::: "vcc", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21");      // a:2 b:3, acc 0,1
    else if constexpr (CASE == 1) asm volatile(MAD4("v2", "v6") MAD4("v3", "v7") ::: "vcc", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21"); // a,b same bank
    else if constexpr (CASE == 2) asm volatile(MAD4("v4", "v3") MAD4("v4", "v7") ::: "vcc", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21"); // a in acc.lo's bank
    else if constexpr (CASE == 3) asm volatile(MAD4("v4", "v5") MAD4("v4", "v5") ::: "vcc", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21"); // a, b in acc.lo / acc.hi banks
    else asm volatile(MAD4("v4", "v4") MAD4("v4", "v4") ::: "vcc", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21");                          // a = b = acc.lo's bank
  }
  uint32_t r;
  asm volatile("v_xor_b32 %0, v8, v12\n\tv_xor_b32 %0, %0, v16\n\tv_xor_b32 %0, %0, v20" : "=v"(r) :: "v8", "v12", "v16", "v20");
  if (r == 0x12345678u) out[t] = r;
}
template <int CASE> void run(uint32_t* d, const char* name, int bpc, int ncu) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CASE>, dim3(ncu * bpc), dim3(256), 0, 0, d, 1u); (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; r++) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k<CASE>, dim3(ncu * bpc), dim3(256), 0, 0, d, 2u + r); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  printf("  %-52s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD\n", name, bpc, best * 1e-3 * 2.4e9 / ((double)bpc * ITERS * 8));
}
int main() {
  hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  uint32_t* d; (void)hipMalloc(&d, 4 * 256 * 8 * p.multiProcessorCount);
  for (int bpc : {2, 8}) {
    run<0>(d, "a, b in the two banks the accumulator does not use", bpc, p.multiProcessorCount);
    run<1>(d, "a and b in one bank", bpc, p.multiProcessorCount);
    run<2>(d, "a in the accumulator's low-dword bank", bpc, p.multiProcessorCount);
    run<3>(d, "a, b in the accumulator's two banks", bpc, p.multiProcessorCount);
    run<4>(d, "a = b = one register in the accumulator's bank", bpc, p.multiProcessorCount);
  }
  return 0;
}
