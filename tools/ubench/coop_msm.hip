// Wave-cooperative multiscalar multiplication against the per-lane Straus kernel, measured (SURVEY §7 "implement
// both ... keep whichever measures faster").  The job is shaped like the largest relation of the path, the third commitment of the
// issuance proof (/root/reference/src/nizk/issuance.rs:119-126: n + 3 terms; C5: 19): here T = 16 variable-base terms per MSM,
// per-item scalars, signed 4-bit windows, 253-bit scalars.
//   A  per-lane Straus (the engine's design): one MSM per lane, 16 window tables per lane in HBM, shared doubling chain.
//   B  wave-cooperative: one MSM per 16-lane slice, lane t owns term t and its window table; per window every lane fetches its
//      entry, the 16 entries are summed with a 4-round butterfly over ds_bpermute (__shfl_xor; every lane ends up with the sum),
//      and the accumulator (replicated in the 16 lanes) takes 4 doublings + 1 addition.
// Both variants produce the compressed result; the host compares them byte for byte before timing is reported.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/coop_msm.hip -o variants/coop_msm      Run: variants/coop_msm [log2 MSMs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../aeonflux_amd/csrc/ge.cuh"
#include "../../aeonflux_amd/csrc/sc.cuh"

constexpr int T = 16;

__device__ __forceinline__ void p3_store(int32_t* p, const ge_p3& q) {
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) { p[l] = q.X.v[l]; p[AFX_FE_LIMBS + l] = q.Y.v[l]; p[2 * AFX_FE_LIMBS + l] = q.Z.v[l]; p[3 * AFX_FE_LIMBS + l] = q.T.v[l]; }
}
__device__ __forceinline__ ge_p3 p3_load(const int32_t* p) {
  ge_p3 q;
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) { q.X.v[l] = p[l]; q.Y.v[l] = p[AFX_FE_LIMBS + l]; q.Z.v[l] = p[2 * AFX_FE_LIMBS + l]; q.T.v[l] = p[3 * AFX_FE_LIMBS + l]; }
  return q;
}
// entry = cached form, 40 dwords (this benchmark does not pack to 128 B: both variants pay the same)
__device__ __forceinline__ void cached_store40(int32_t* p, const ge_cached& q) {
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) { p[l] = q.YpX.v[l]; p[AFX_FE_LIMBS + l] = q.YmX.v[l]; p[2 * AFX_FE_LIMBS + l] = q.Z2.v[l]; p[3 * AFX_FE_LIMBS + l] = q.T2d.v[l]; }
}
__device__ __forceinline__ ge_cached cached_load40(const int32_t* p) {
  ge_cached q;
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) { q.YpX.v[l] = p[l]; q.YmX.v[l] = p[AFX_FE_LIMBS + l]; q.Z2.v[l] = p[2 * AFX_FE_LIMBS + l]; q.T2d.v[l] = p[3 * AFX_FE_LIMBS + l]; }
  return q;
}

// inputs: point t of MSM m = from_uniform of a counter-derived 64-byte string; scalar = 253 pseudo-random bits
__global__ void k_inputs(int32_t* pts, uint32_t* scalars, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * T) return;
  uint32_t w[16], x = i * 2654435761u + 12345u;
  for (int k = 0; k < 16; k++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w[k] = x; }
  p3_store(pts + (size_t)i * 40, ge_carry(ristretto_from_uniform(w)));
  for (int k = 0; k < 8; k++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; scalars[(size_t)i * 8 + k] = x; }
  scalars[(size_t)i * 8 + 7] &= 0x0fffffffu;
}

// window table of one base: entries 0..8 in cached form
__device__ __forceinline__ void build_table(int32_t* tab, const ge_p3& P) {
  const ge_cached cP = ge_p3_to_cached_reduced(P);
  cached_store40(tab, ge_cached_identity());
  cached_store40(tab + 40, cP);
  ge_p3 Q = P;
#pragma unroll 1
  for (int k = 2; k < 9; k++) {
    Q = ge_p1p1_to_p3(ge_add_cached(Q, cP, false));
    cached_store40(tab + 40 * k, ge_p3_to_cached_reduced(Q));
  }
}
__device__ __forceinline__ int digit_of(const uint32_t b[8], int w) { return (int)((b[w >> 3] >> ((w & 7) * 4)) & 15u) - 8; }

// A: one MSM per lane.  item i: terms t = 0..T-1 at index t * n + i
__global__ void __launch_bounds__(256, 2) k_straus(const int32_t* __restrict__ pts, const uint32_t* __restrict__ scalars, int32_t* __restrict__ tabs,
                                                   uint32_t* __restrict__ digits, uint8_t* __restrict__ out, uint32_t n) {
  const uint32_t i = min(blockIdx.x * 256 + threadIdx.x, n - 1);
#pragma unroll 1
  for (int t = 0; t < T; t++) {
    build_table(tabs + ((size_t)t * n + i) * 360, p3_load(pts + ((size_t)i * T + t) * 40));
    sc s;
    for (int k = 0; k < 8; k++) s.v[k] = scalars[((size_t)i * T + t) * 8 + k];
    uint32_t b[8];
    sc_bias(b, s, 0x88888888u);
    for (int k = 0; k < 8; k++) digits[((size_t)t * 8 + k) * n + i] = b[k];
  }
  ge_p3 acc = ge_identity();
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {
    if (w != 63) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) acc = ge_double(acc);
    }
#pragma unroll 1
    for (int t = 0; t < T; t++) {
      const uint32_t word = digits[((size_t)t * 8 + (w >> 3)) * n + i];
      const int d = (int)((word >> ((w & 7) * 4)) & 15u) - 8;
      const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
      acc = ge_p1p1_to_p3(ge_add_cached(acc, cached_load40(tabs + ((size_t)t * n + i) * 360 + idx * 40), d < 0));
    }
  }
  uint32_t e[8];
  ristretto_encode(e, acc);
  for (int k = 0; k < 8; k++) reinterpret_cast<uint32_t*>(out)[(size_t)i * 8 + k] = e[k];
}

__device__ __forceinline__ ge_p3 shfl_xor_p3(const ge_p3& p, int mask) {
  ge_p3 r;
#pragma unroll
  for (int l = 0; l < AFX_FE_LIMBS; l++) {
    r.X.v[l] = __shfl_xor(p.X.v[l], mask, 16); r.Y.v[l] = __shfl_xor(p.Y.v[l], mask, 16);
    r.Z.v[l] = __shfl_xor(p.Z.v[l], mask, 16); r.T.v[l] = __shfl_xor(p.T.v[l], mask, 16);
  }
  return r;
}

// B: one MSM per 16-lane slice; lane t of the slice owns term t
__global__ void __launch_bounds__(256, 2) k_coop(const int32_t* __restrict__ pts, const uint32_t* __restrict__ scalars, int32_t* __restrict__ tabs,
                                                 uint8_t* __restrict__ out, uint32_t n) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  const uint32_t m = min(g / T, n - 1), t = g % T;
  const ge_p3 P = p3_load(pts + ((size_t)m * T + t) * 40);
  int32_t* tab = tabs + ((size_t)m * T + t) * 360;
  build_table(tab, P);
  sc s;
  for (int k = 0; k < 8; k++) s.v[k] = scalars[((size_t)m * T + t) * 8 + k];
  uint32_t b[8];
  sc_bias(b, s, 0x88888888u);
  ge_p3 acc = ge_identity();
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {
    // this lane's term: digit * P as an extended point (entry d of its own table, negated when the digit is negative)
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) if ((w >> 3) == k) d = (int)((b[k] >> ((w & 7) * 4)) & 15u) - 8;
    const uint32_t idx = (uint32_t)(d < 0 ? -d : d);
    ge_p3 mine = ge_p1p1_to_p3(ge_add_cached(ge_identity(), cached_load40(tab + idx * 40), d < 0));
    // butterfly all-reduce over the 16 lanes of the slice: 4 rounds, every lane ends with the window's sum
#pragma unroll 1
    for (int mask = 8; mask >= 1; mask >>= 1) mine = ge_add(mine, shfl_xor_p3(mine, mask));
    if (w != 63) {
#pragma unroll 1
      for (int k = 0; k < 4; k++) acc = ge_double(acc);
    }
    acc = ge_add(acc, mine);
  }
  uint32_t e[8];
  ristretto_encode(e, acc);
  if (t == 0 && g / T < n)
    for (int k = 0; k < 8; k++) reinterpret_cast<uint32_t*>(out)[(size_t)m * 8 + k] = e[k];
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 16;
  const uint32_t n = 1u << lg;
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { printf("no device\n"); return 1; }
  int32_t *pts, *tabs; uint32_t *scalars, *digits; uint8_t *outA, *outB;
  hipMalloc(&pts, (size_t)n * T * 160); hipMalloc(&scalars, (size_t)n * T * 32); hipMalloc(&tabs, (size_t)n * T * 1440);
  hipMalloc(&digits, (size_t)n * T * 32); hipMalloc(&outA, (size_t)n * 32); hipMalloc(&outB, (size_t)n * 32);
  hipLaunchKernelGGL(k_inputs, dim3((n * T + 255) / 256), dim3(256), 0, 0, pts, scalars, n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float msA = 1e30f, msB = 1e30f;
  for (int r = 0; r < 3; r++) {
    float ms;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_straus, dim3((n + 255) / 256), dim3(256), 0, 0, pts, scalars, tabs, digits, outA, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (ms < msA) msA = ms;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_coop, dim3(((size_t)n * T + 255) / 256), dim3(256), 0, 0, pts, scalars, tabs, outB, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (ms < msB) msB = ms;
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<uint8_t> a((size_t)n * 32), b((size_t)n * 32);
  hipMemcpy(a.data(), outA, a.size(), hipMemcpyDeviceToHost); hipMemcpy(b.data(), outB, b.size(), hipMemcpyDeviceToHost);
  size_t diff = 0, zero = 0;
  for (uint32_t i = 0; i < n; i++) { diff += memcmp(&a[(size_t)i * 32], &b[(size_t)i * 32], 32) != 0; bool z = true; for (int k = 0; k < 32; k++) z &= a[(size_t)i * 32 + k] == 0; zero += z; }
  printf("# %d-term MSMs, %u of them, %s (%d CUs)\n", T, n, pr.name, pr.multiProcessorCount);
  printf("results: %zu of %u differ between the two variants, %zu are the identity (expected 0 and 0)\n", diff, n, zero);
  printf("A per-lane Straus      %9.3f ms   %8.3f M MSMs/s\n", msA, n / msA / 1e3);
  printf("B wave-cooperative     %9.3f ms   %8.3f M MSMs/s   (B/A time = %.2f)\n", msB, n / msB / 1e3, msB / msA);
  return diff != 0;
}
