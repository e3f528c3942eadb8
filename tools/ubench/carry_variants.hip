// Which part of the centred carry step costs what on gfx950?  fe_mul with the carry step varied:
//   0 raw (floor carry, masked limb)            1 shipped centred form (2^50 into the high dword, limb - 2^(b-1))
//   2 floor carry, limb - 2^(b-1) only          3 2^50 into the high dword only (limb masked)
//   4 rounding constant added by a mad (H = 1 * R + H), limb - 2^(b-1)
//   5 floor carries on the serial path, limbs centred afterwards with 32-bit operations off that path
// Results are not all meaningful field products; only the instruction mix matters.
// Build: hipcc -O3 --offload-arch=gfx950 carry_variants.hip -o build/carry_variants
#include "fe10_old.cuh"   /* the 10 x 25.5-bit field arithmetic of round 1, frozen for these measurements */
#include <cstdio>
#include <vector>
constexpr int ITERS = 2000;
template <int V>
AFX_DEV fe mul_v(const fe& f, const fe& g) {
  int32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) { g19[i] = (int32_t)(19u * (uint32_t)g.v[i]); f2[i] = (int32_t)(2u * (uint32_t)f.v[i]); }
  fe r;
  int64_t c = (V == 1) ? (1LL << 25) : 0;
  uint32_t u0 = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    int64_t H = c;
    const int bits = (k & 1) ? 25 : 26;
    if (V == 4) { H += (int64_t)1 * (int64_t)(1 << (bits - 1)); AFX_PIN(H); }
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      const int32_t a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
      const int32_t b = (i > k) ? g19[j] : g.v[j];
      H += (int64_t)a * (int64_t)b;
      AFX_PIN(H);
    }
    const uint32_t lo = (uint32_t)H & ((1u << bits) - 1);
    const bool sub = V == 1 || V == 2 || V == 4;
    if (k == 0) u0 = lo; else r.v[k] = sub ? (int32_t)lo - (1 << (bits - 1)) : (int32_t)lo;
    const bool hi = (V == 1 || V == 3) && k < 9;
    c = hi ? ((H + (1LL << 50)) >> bits) : (H >> bits);
  }
  int64_t H0 = (int64_t)u0 + c * 19;
  r.v[0] = (int32_t)((uint32_t)H0 & 0x3ffffffu);
  r.v[1] += (int32_t)(H0 >> 26);
  if (V == 5) {
    // post-hoc centring: t_k = top bit of limb k; limb k -= t_k << b_k; limb k+1 += t_k (limb 0 += 19 t_9)
    int32_t t[10];
#pragma unroll
    for (int k = 0; k < 10; k++) { const int bits = (k & 1) ? 25 : 26; t[k] = (int32_t)((uint32_t)r.v[k] >> (bits - 1)) & 1; }
#pragma unroll
    for (int k = 0; k < 10; k++) {
      const int bits = (k & 1) ? 25 : 26;
      r.v[k] = r.v[k] - (t[k] << bits) + (k == 0 ? 19 * t[9] : t[k - 1]);
    }
  }
  return r;
}
template <int V>
__global__ void __launch_bounds__(256, 2) k(int32_t* p) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  fe a, b;
  for (int i = 0; i < 10; i++) { a.v[i] = p[i * 512 * 256 + t] & 0x1ffffff; b.v[i] = (p[i * 512 * 256 + t] >> 3) & 0xffffff; }
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) a = mul_v<V>(a, b);
  for (int i = 0; i < 10; i++) p[i * 512 * 256 + t] = a.v[i];
}
template <int V> void run(int32_t* d, const char* name, int ncu) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(ncu * 2), dim3(256), 0, 0, d); (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; r++) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(ncu * 2), dim3(256), 0, 0, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  printf("  %-64s %.0f cycles per wave-level multiplication per SIMD\n", name, best * 1e-3 * 2.4e9 / (2.0 * ITERS));
}
int main() {
  hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, 0) != hipSuccess) return 1;
  int32_t* d; (void)hipMalloc(&d, sizeof(int32_t) * 10 * 512 * 256);
  std::vector<int32_t> h(10 * 512 * 256);
  for (size_t i = 0; i < h.size(); i++) h[i] = (int32_t)(i * 2654435761u >> 4);
  (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0>(d, "0 raw: floor carry, masked limb", pr.multiProcessorCount);
  run<1>(d, "1 centred as shipped: +2^50 on the high dword, limb - 2^(b-1)", pr.multiProcessorCount);
  run<2>(d, "2 floor carry, limb - 2^(b-1)", pr.multiProcessorCount);
  run<3>(d, "3 +2^50 on the high dword, masked limb", pr.multiProcessorCount);
  run<4>(d, "4 rounding constant through a mad, limb - 2^(b-1)", pr.multiProcessorCount);
  run<5>(d, "5 floor carries, limbs centred afterwards (32-bit, off the serial path)", pr.multiProcessorCount);
  return 0;
}
